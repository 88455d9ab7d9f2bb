#include "spartan.h"
#include "snark.h"
#include "pool.h"
#include <stdio.h>
#include <x86intrin.h>
#include <chrono>
using namespace otti;
extern unsigned long long g_tm[16];
int main() {
    auto g = gens_new(16, 16, 1);
    uint8_t w[64] = {1,2,3}; Fr s = fr_from_bytes_wide(w); w[5] = 9; Fr s2 = fr_from_bytes_wide(w); w[7] = 3; const Pt rnd = pt_from_uniform_bytes(w);
    SpinPool::Session session; SpinPool &pool = SpinPool::get();
    const int rounds = 2000;
    Transcript tr("bench", 5); RandomTape tape(w);
    SumcheckState st; sumcheck_draw_tape(st, tape, rounds, 4);
    for (auto &p : st.pre) { Term t = {g->sc_4.h, st.blinds_poly[0]}; p.bp_h = g->commit_terms(&t, 1); p.be_h = p.bp_h; p.rb_h = p.bp_h; p.delta = p.bp_h; p.to_fe(); pt_encode(p.delta_c.b, p.delta); }
    st.claim = s; st.blind_claim = s2; pt_encode(st.comm_claim.b, rnd);
    ZKSumcheckProof pf; pf.comm_polys.resize(rounds); pf.comm_evals.resize(rounds); pf.proofs.resize(rounds);
    auto t0 = std::chrono::steady_clock::now(); unsigned long long c0 = __rdtsc();
    for (int j = 0; j < rounds; j++) {
        Fr ev[4] = {s, fr_sub(st.claim, s), s2, fr_mul(s, s2)};
        RoundPart1 p1 = sumcheck_round_begin(pf, j, ev, 4, st, *g, g->sc_4, tr);
        sumcheck_round_finish(pf, j, p1, st, *g, g->sc_4, tr);
        s = fr_add(s, p1.r_j);
    }
    unsigned long long c1 = __rdtsc(); double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    double ns_per_tick = ns / (double)(c1 - c0);
    const char *names[15] = {"finish: eval + fixed-base(eval)", "finish: add + compress comm_eval", "finish: 2 appends + 2 challenges", "finish: target/blind/a/ad scalars", "finish: submit two tasks", "finish: own fixed-base (cy_g)", "finish: wait helper 0 (cy_h)", "finish: add + compress Cy", "finish: wait helper 1 (beta)", "finish: protocol name + 5 appends + a + challenge", "finish: z vectors, copies", "begin: unipoly", "begin: 4 fixed-base in parallel", "begin: 4 adds + compress", "begin: append + challenge"};
    double tot = 0;
    for (int k = 0; k < 15; k++) { double v = g_tm[k] * ns_per_tick / rounds; tot += v; printf("%-52s %8.0f ns\n", names[k], v); }
    printf("%-52s %8.0f ns (wall per round %.0f ns, %d threads)\n", "sum", tot, ns / rounds, pool.workers() + 1);
}
