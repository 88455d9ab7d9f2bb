# SNARK::prove at 2^20 with the host's last sum-check rounds in the AVX-512 IFMA form (default) / the scalar form, and with the host taking
# more or fewer rounds of every layer from the persistent launch (OTTI_PC_LGT_MANY / _FEW: log2 of the host tail's table length for
# batches of >= 8 / fewer instances).
python - <<'PY'
import otti_amd as oa
print("host_tail_bench, microseconds per layer:")
for (np_, nd, T) in [(12, 6, 16), (12, 6, 32), (12, 6, 64), (12, 0, 16), (12, 0, 32), (12, 0, 64), (4, 0, 32), (4, 0, 64), (4, 0, 128), (4, 0, 256)]:
    print(f"  np={np_} nd={nd} T={T}: " + "; ".join(f"{th} thread(s) " + ", ".join(f"{k} {v:.1f}" for k, v in oa.host_tail_bench(np_, nd, T, th, 300).items()) for th in (1, 2, 4)))
PY
for v in "default" "OTTI_HOST_FR8=0" "OTTI_PC_LGT_MANY=4 OTTI_PC_LGT_FEW=5" "OTTI_PC_LGT_MANY=5 OTTI_PC_LGT_FEW=6" "OTTI_PC_LGT_MANY=5 OTTI_PC_LGT_FEW=7" "OTTI_PC_LGT_MANY=5 OTTI_PC_LGT_FEW=8" "OTTI_PC_LGT_MANY=6 OTTI_PC_LGT_FEW=7" "OTTI_PC_LGT_MANY=6 OTTI_PC_LGT_FEW=7 OTTI_HOST_TAIL_GRAIN=24"; do
  echo "=== $v"
  if [ "$v" = default ]; then OTTI_TRACE=1 python tools/snark_probe.py 20 5 2>&1 | grep "  prove\|pcbatch" | tail -6 | cut -c1-420; else env $v OTTI_TRACE=1 python tools/snark_probe.py 20 5 2>&1 | grep "  prove\|pcbatch\|rror" | tail -6 | cut -c1-420; fi
done
