"""Where one zero-knowledge sum-check round's host work goes (no GPU needed): makes an instrumented copy of
otti_amd/csrc/spartan_host.cpp (cycle counters between the steps of sumcheck_round_begin / sumcheck_round_finish) in a scratch
directory, builds it with the product's host compiler and flags beside tools/roundbench/main.cpp, and runs 2000 rounds.
usage: python tools/roundbench/run.py [threads]     (threads = OTTI_HOST_THREADS, default 4)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = os.path.join(ROOT, "otti_amd", "csrc")
src = open(os.path.join(S, "spartan_host.cpp")).read()


def rep(old, new):
    global src
    assert old in src, "spartan_host.cpp changed: update tools/roundbench/run.py (%r)" % old[:60]
    src = src.replace(old, new, 1)


rep('#include "spartan.h"', '#include "spartan.h"\n#include <x86intrin.h>\nunsigned long long g_tm[16];\n#define TM(k) do { unsigned long long t_ = __rdtsc(); g_tm[k] += t_ - t_last; t_last = t_; } while (0)\n')
rep("    const size_t ne = p1.ne; const RoundPre &pre = st.pre[j]; (void)gn;\n    Fr eval = unipoly_eval(p1.poly, ne, p1.r_j);",
    "    unsigned long long t_last = __rdtsc();\n    const size_t ne = p1.ne; const RoundPre &pre = st.pre[j]; (void)gn;\n    Fr eval = unipoly_eval(p1.poly, ne, p1.r_j);")
rep("    CPoint comm_eval = encode_sum(g.commit_terms_fe(&te, 1), pre.be_fe);", "    PtFe ce_ = g.commit_terms_fe(&te, 1); TM(0);\n    CPoint comm_eval = encode_sum(ce_, pre.be_fe); TM(1);\n    //")
rep('    std::vector<Fr> w = tr.challenge_vector("combine_two_claims_to_one", 2);\n    Fr target =', '    std::vector<Fr> w = tr.challenge_vector("combine_two_claims_to_one", 2); TM(2);\n    Fr target =')
rep("    DotProductProof dp; PtFe cy_g, cy_h; CPoint Cy;\n", "    TM(3);\n    DotProductProof dp; PtFe cy_g, cy_h; CPoint Cy;\n")
rep("        pool.submit(0, tasks[1]); pool.submit(1, tasks[2]);\n        tasks[0](); pool.wait(0);\n", "        pool.submit(0, tasks[1]); pool.submit(1, tasks[2]); TM(4);\n        tasks[0](); TM(5); pool.wait(0); TM(6);\n")
rep("        pool.wait(1);\n", "        TM(7); pool.wait(1); TM(8);\n")
rep('    Fr c = tr.challenge_scalar("c");\n    dp.z.resize(ne);', '    Fr c = tr.challenge_scalar("c"); TM(9);\n    dp.z.resize(ne);')
rep("    st.claim = eval; st.comm_claim = comm_eval; pf.comm_evals[j] = comm_eval;\n}", "    st.claim = eval; st.comm_claim = comm_eval; pf.comm_evals[j] = comm_eval; TM(10);\n}")
rep("    RoundPart1 p; p.ne = ne;\n    unipoly_from_evals(p.poly, evals, ne);", "    unsigned long long t_last = __rdtsc();\n    RoundPart1 p; p.ne = ne;\n    unipoly_from_evals(p.poly, evals, ne); TM(11);")
rep("    SpinPool::get().parallel(tasks, (int)ne);\n    PtFe sum = st.pre[j].bp_fe;", "    SpinPool::get().parallel(tasks, (int)ne); TM(12);\n    PtFe sum = st.pre[j].bp_fe;")
rep('    pt_encode_fe(pf.comm_polys[j].b, sum);\n    tr.append_point("comm_poly", pf.comm_polys[j].b);\n    p.r_j = tr.challenge_scalar("challenge_nextround");\n    return p;',
    '    pt_encode_fe(pf.comm_polys[j].b, sum); TM(13);\n    tr.append_point("comm_poly", pf.comm_polys[j].b);\n    p.r_j = tr.challenge_scalar("challenge_nextround"); TM(14);\n    return p;')

with tempfile.TemporaryDirectory() as d:
    timed = os.path.join(d, "spartan_host_timed.cpp")
    open(timed, "w").write(src)
    exe = os.path.join(d, "roundbench")
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
    subprocess.check_call([cxx, "-O3", "-std=c++17", "-march=x86-64-v3", "-I" + S, "-Wno-unused-result", os.path.join(ROOT, "tools", "roundbench", "main.cpp"), timed] +
                          [os.path.join(S, f) for f in ("hash.cpp", "hostfast.cpp", "hostifma.cpp", "hostgroup.cpp", "snark_host.cpp")] + ["-o", exe, "-lpthread"])
    env = dict(os.environ); env["OTTI_HOST_THREADS"] = sys.argv[1] if len(sys.argv) > 1 else "4"
    for _ in range(3):
        subprocess.check_call([exe], env=env)
        print()
