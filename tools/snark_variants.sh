for v in "default" "OTTI_DEREFS_AHEAD=0" "OTTI_DEREFS_FREE_CUS=128" "OTTI_DEREFS_FREE_CUS=32" "OTTI_PC_TAIL=0"; do
  echo "=== $v"
  if [ "$v" = default ]; then python tools/snark_probe.py 20 6 2>&1 | grep "prove" | tail -4; else env $v python tools/snark_probe.py 20 6 2>&1 | grep "prove" | tail -4; fi
done
