#!/bin/bash
# How much aggregate throughput do B independent prover processes (each with T prover threads) get out of ONE GPU?
# usage: concurrent_probe.sh B [window [T]]
B=$1; W=${2:-12}
for i in $(seq 1 $B); do
  OTTI_MSM_WINDOW=$W python3 bench.py --no-cpu-baseline --steps 100 --warmup 3 --concurrent ${3:-1} > gpurun_out/cc_${B}_$i.json 2> gpurun_out/cc_${B}_$i.err &
done
wait
python3 - <<PY
import json, glob
tot = 0
for f in sorted(glob.glob("gpurun_out/cc_${B}_*.json")):
    d = json.loads(open(f).read()); tot += d["value"]; print(f, d["ms_per_step"], round(d["value"] / 1e6, 1))
print("B=${B} aggregate M constraints/s:", round(tot / 1e6, 1))
PY
