for v in "OTTI_GO_POLLERS=1" "OTTI_GO_POLLERS=4" "OTTI_GO_POLLERS=1 OTTI_RELAY=0" "OTTI_GO_POLLERS=2"; do
  for lg in 18 20; do
    echo "=== $v 2^$lg"
    env $v python bench.py --log2-constraints $lg --steps 60 --warmup 5 --no-snark --no-sweep --no-e2e --in-flight -1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_p50'], d['ms_per_step_p99'], d['ms_per_step_max'], d['stage_ms'])"
  done
done
