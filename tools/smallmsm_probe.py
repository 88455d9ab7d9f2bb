"""Device time of the small (one/two-row) fixed-base MSM launches through the kernel-level C ABI: one bullet-reduction round per
launch at R = 1024 (what a 2^20 proof runs ten times), HIP events around the launch alone."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import otti_amd as oa
import orc
K = oa.kernels
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 1 << lg; V = 1 << (2 * lg)
rng = np.random.default_rng(1)
gens = oa.NIZKGens.new(V, V, 1)
a, b = orc.rand_fr(rng, n), orc.rand_fr(rng, n); s = np.repeat(orc.fr_from_ints([1]), n, axis=0)
bl = orc.rand_fr(rng, 2); u = orc.rand_fr(rng, 1); ui = orc.fr_from_ints([pow(orc.fr_to_ints(u)[0], -1, orc.L_ORDER)])
K.bullet_round(gens, n, a, b, s, bl)
print("window bits / table bytes:", gens.table_info)
cur, ca, cb, cs = n, a, b, s
first = True
while cur >= 2:
    best = 1e9
    for _ in range(5):
        LR, na, nb, ns, ms = K.bullet_round(gens, cur, ca, cb, cs, bl, None if first else u, None if first else ui)
        best = min(best, ms)
    print(f"bullet round n_cur={cur:5d} fold={not first}: {best*1e3:8.1f} us")
    ca, cb, cs, cur, first = na, nb, ns, cur // 2, False
Z, blind = orc.rand_fr(rng, n), orc.rand_fr(rng, 1)
best = min(K.msm_rows(gens, Z, 1, n, blind)[1] for _ in range(5))
print(f"one-row commitment (Cx / delta shape), {n} terms: {best*1e3:8.1f} us")
