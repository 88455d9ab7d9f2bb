"""Host-side primitives of the sequential path on this machine (no GPU needed): otti_host_microbench, nanoseconds per operation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa
oa.host_selftest(100)
best = None
for _ in range(5):
    d = oa.host_microbench()
    best = d if best is None else {k: min(best[k], v) for k, v in d.items()}
for k, v in best.items():
    print("%-26s %10.1f" % (k, v))
