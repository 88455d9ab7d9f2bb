# SNARK::prove at 2^20: elements per workgroup of the persistent tail at its start (OTTI_PC_TAIL_PER_WG), host tail lengths
for v in "OTTI_PC_TAIL_PER_WG=128" "OTTI_PC_TAIL_PER_WG=256" "OTTI_PC_TAIL_PER_WG=512" "OTTI_PC_TAIL_PER_WG=1024" "OTTI_PC_TAIL_PER_WG=256 OTTI_PC_LGT_MANY=5 OTTI_PC_LGT_FEW=6" "OTTI_PC_TAIL_PER_WG=512 OTTI_PC_LGT_MANY=5 OTTI_PC_LGT_FEW=6" "OTTI_PC_TAIL_PER_WG=512 OTTI_PC_LGT_MANY=6 OTTI_PC_LGT_FEW=7" "OTTI_PC_TAIL_PER_WG=128 OTTI_PC_LGT_MANY=5 OTTI_PC_LGT_FEW=6"; do
  echo "=== $v"
  env $v OTTI_TRACE=1 python tools/snark_probe.py 20 5 2>&1 | grep "  prove\|pcbatch\|rror" | tail -6 | cut -c1-420
done
