# the six-proofs-in-flight leg of bench.py: commitment MSMs on the CU-masked stream (default) or not, helper threads per prover, proofs in flight
F="--steps 20 --warmup 1 --no-snark --no-sweep --no-e2e --no-cpu-baseline"
run() { # label, env assignment(s), in-flight count
  L=$(env $2 python bench.py $F --in-flight $3 2>/dev/null | tail -1)
  echo "$1 (in flight $3): $(echo "$L" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('single', d['ms_per_step'], 'ms; in_flight', round(d['in_flight']['value']/1e6,1), 'M/s, latency', d['in_flight']['latency_ms_per_proof'])")"
}
for rep in 1 2 3; do
  run "default" "X=1" 6
  run "OTTI_INFLIGHT_MASK=1" "OTTI_INFLIGHT_MASK=1" 6
  run "OTTI_HOST_THREADS_FIXED=1" "OTTI_HOST_THREADS_FIXED=1" 6
done
run "OTTI_HOST_THREADS=2" "OTTI_HOST_THREADS=2" 6
run "OTTI_HOST_THREADS=2" "OTTI_HOST_THREADS=2" 8
run "default" "X=1" 4
run "default" "X=1" 8
