"""Device time of the bulk fixed-base MSM (the witness-commitment launch, k_msm_rows<0>) through the kernel-level C ABI."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import otti_amd as oa
import orc
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ell = lg; L = 1 << (ell // 2); R = 1 << (ell - ell // 2)
rng = np.random.default_rng(1)
gens = oa.NIZKGens.new(1 << lg, 1 << lg, 1)
Z = orc.rand_fr(rng, L * R); bl = orc.rand_fr(rng, L)
oa.kernels.msm_rows(gens, Z, L, R, bl)
print("window bits / table bytes:", gens.table_info)
ts = sorted(oa.kernels.msm_rows(gens, Z, L, R, bl)[1] for _ in range(7))
print(f"2^{lg}: {L} rows x {R} terms: min {ts[0]*1e3:.1f} us  median {ts[3]*1e3:.1f} us  (includes the blind row and the finish launch)")
