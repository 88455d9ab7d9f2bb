"""Time of otti_prepare_device (window-table build) for a 2^lg instance.  usage: prep_time.py [lg]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
r = oa.synth_r1cs(1 << 10, 10, 1)
inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
g0 = oa.NIZKGens.new(1 << 10, 1 << 10, 10)
inst.prepare_device(g0)                                   # context creation etc. out of the way
t0 = time.perf_counter(); gens = oa.NIZKGens.new(n, n, 10); t1 = time.perf_counter()
t2 = time.perf_counter(); oa.lib.otti_prepare_device(None, gens._h); t3 = time.perf_counter()
print("gens_new %.1f ms, table build %.1f ms, info %s" % (1e3 * (t1 - t0), 1e3 * (t3 - t2), gens.table_info))
