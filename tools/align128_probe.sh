# 96-byte against 128-byte (line-aligned, padded) window-table entries: the library is built twice (the second time with
# -DOTTI_TABLE_ALIGN128 into _exp128/, see tools/README.md) and bench.py's timed region (20 proofs) runs alternately on both.
# usage on the GPU box, from the repository root: bash tools/align128_probe.sh
set -e
F="--no-snark --no-e2e --no-sweep --in-flight -1 --no-cpu-baseline"
cp otti_amd/libottispartan.so /tmp/lib96.so
for rep in 1 2 3; do
  for v in 96 128; do
    if [ $v = 96 ]; then cp /tmp/lib96.so otti_amd/libottispartan.so; else cp _exp128/libottispartan.so otti_amd/libottispartan.so; fi
    python3 bench.py $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('entries of $v bytes: table %.1f GB, k_msm_rows<0> %.4f ms (HIP events, %d launches), NIZK::prove %.3f ms' % (d['config']['msm_table_GB'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['ms_per_step']))"
  done
done
cp /tmp/lib96.so otti_amd/libottispartan.so
