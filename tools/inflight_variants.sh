# the six-proofs-in-flight leg of bench.py with the commitment MSMs on the CU-masked stream (default), without the mask, and with other splits
F="--steps 5 --warmup 1 --in-flight 6 --no-snark --no-sweep --no-e2e --no-cpu-baseline"
for v in "default" "OTTI_INFLIGHT_MASK=0" "OTTI_DEREFS_FREE_CUS=32" "OTTI_DEREFS_FREE_CUS=96" "default" "OTTI_INFLIGHT_MASK=0"; do
  if [ "$v" = default ]; then L=$(python bench.py $F 2>/dev/null | tail -1); else L=$(env $v python bench.py $F 2>/dev/null | tail -1); fi
  echo "$v: $(echo "$L" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('single', d['ms_per_step'], 'ms; in_flight', round(d['in_flight']['value']/1e6,1), 'M/s, latency', d['in_flight']['latency_ms_per_proof'])")"
done
