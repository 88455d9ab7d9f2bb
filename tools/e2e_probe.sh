#!/bin/bash
# one-shot `spzk verify --nizk` on a synthetic 2^LG triple: wall time of the whole process and its own stage lines
LG=${1:-20}; D=$(mktemp -d); N=$((1<<LG))
./otti_amd/spzk synth $N $D/w > /dev/null
for i in 1 2 3; do
  S=$(date +%s%N); OTTI_TRACE=${OTTI_TRACE_E2E:-} ./otti_amd/spzk verify --nizk $D/w.zkif $D/w.inp.zkif $D/w.wit.zkif --seed $(printf '2a%.0s' {1..32}) > $D/out.txt 2>&1; RC=$?; E=$(date +%s%N)
  echo "run $i: rc=$RC wall $(( (E-S)/1000000 )) ms"; grep -E "zkif_load|setup|host objects|NIZK::prove|NIZK::verify|Verification|otti\]" $D/out.txt | tr '\n' ' '; echo
done
rm -rf $D
