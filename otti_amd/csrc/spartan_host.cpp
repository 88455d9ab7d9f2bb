// Host-side objects of the proving path: instance building, generator derivation, proof (de)serialisation, sigma protocols,
// verifier, synthetic instances.  See spartan.h for what each piece replaces upstream.
#include "spartan.h"
#include "snark.h"
#include "pool.h"
#include <algorithm>
#include <chrono>
#include <numeric>
#include <thread>
#include <functional>
#include <atomic>
#include <system_error>
#include <memory>

namespace otti {

// ================================================================================================ instance (lib.rs Instance::new)
std::unique_ptr<Instance> instance_new(size_t num_cons, size_t num_vars, size_t num_inputs, const otti_entry *A, size_t nA,
                                       const otti_entry *B, size_t nB, const otti_entry *C, size_t nC) {
    size_t nvp = next_pow2(std::max(num_vars, num_inputs + 1));
    size_t ncp = num_cons < 2 ? 2 : next_pow2(num_cons);
    if (ncp > ((size_t)1 << 30) || nvp > ((size_t)1 << 30)) throw Error(OTTI_ERR_BAD_ARG, "instance too large for 32-bit indices");
    auto I = std::make_unique<Instance>();
    I->num_cons = ncp; I->num_vars = nvp; I->num_inputs = num_inputs; I->given_cons = num_cons;
    const otti_entry *src[3] = {A, B, C}; size_t cnt[3] = {nA, nB, nC};
    // validation and conversion to Montgomery form (one field multiplication per entry) in blocks over the host cores; the error a
    // sequential walk would meet first is the one reported
    const unsigned nt = (nA + nB + nC) < 8192 ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    for (int k = 0; k < 3; k++) {
        SparseMat &m = I->M[k];
        m.row.resize(cnt[k]); m.col.resize(cnt[k]); m.val.resize(cnt[k]);
        std::vector<size_t> bad_at(nt, SIZE_MAX); std::vector<int> bad_code(nt, 0);
        auto work = [&](unsigned t) {
            const size_t i0 = cnt[k] * t / nt, i1 = cnt[k] * (t + 1) / nt;
            for (size_t i = i0; i < i1; i++) {
                const otti_entry &e = src[k][i];
                int code = 0;
                if (e.row >= num_cons || e.col >= num_vars + 1 + num_inputs) code = OTTI_ERR_INVALID_INDEX;
                Fr v = fr_zero();
                if (!code && !fr_from_bytes(v, e.val)) code = OTTI_ERR_INVALID_SCALAR;
                if (code) { bad_at[t] = i; bad_code[t] = code; return; }
                // columns >= num_vars reference the constant 1 or an input: shift by the padding of the variable block
                const size_t col = e.col >= num_vars ? e.col + nvp - num_vars : e.col;
                m.row[i] = (uint32_t)e.row; m.col[i] = (uint32_t)col; m.val[i] = v;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < nt; t++) if (bad_at[t] != SIZE_MAX) {
            const otti_entry &e = src[k][bad_at[t]];
            if (e.row >= num_cons) throw Error(OTTI_ERR_INVALID_INDEX, "row index out of range");
            if (e.col >= num_vars + 1 + num_inputs) throw Error(OTTI_ERR_INVALID_INDEX, "column index out of range");
            throw Error(OTTI_ERR_INVALID_SCALAR, "non-canonical scalar in matrix");
        }
    }
    // (upstream also appends explicit zero entries when num_cons < 2; zeros change nothing for the satisfiability proof and are not stored)
    // the access paths the kernels read (CSR by row and by column) are built on the device from these lists (k_sparse.hip upload_instance)
    return I;
}

std::vector<Fr> eq_evals_host(const Fr *r, size_t ell) {
    std::vector<Fr> ev((size_t)1 << ell);
    ev[0] = fr_one();
    size_t size = 1;
    for (size_t j = 0; j < ell; j++) {
        for (size_t k = size; k-- > 0;) { Fr hi = fr_mul(ev[k], r[j]); ev[2 * k] = fr_sub(ev[k], hi); ev[2 * k + 1] = hi; }
        size *= 2;
    }
    return ev;
}

void Instance::evaluate(const std::vector<Fr> &rx, const std::vector<Fr> &ry, Fr out[3]) const {
    std::vector<Fr> ex = eq_evals_host(rx.data(), rx.size()), ey = eq_evals_host(ry.data(), ry.size());
    for (int k = 0; k < 3; k++) {
        Fr acc = fr_zero();
        const SparseMat &m = M[k];
        for (size_t i = 0; i < m.val.size(); i++) acc = fr_add(acc, fr_mul(fr_mul(ex[m.row[i]], ey[m.col[i]]), m.val[i]));
        out[k] = acc;
    }
}

static std::vector<Fr> build_z(const Instance &I, const std::vector<Fr> &vars, const std::vector<Fr> &inputs) {
    std::vector<Fr> z(2 * I.num_vars, fr_zero());
    std::copy(vars.begin(), vars.end(), z.begin());
    z[I.num_vars] = fr_one();
    std::copy(inputs.begin(), inputs.end(), z.begin() + I.num_vars + 1);
    return z;
}

bool Instance::is_sat(const std::vector<Fr> &vars, const std::vector<Fr> &inputs) const {
    std::vector<Fr> z = build_z(*this, vars, inputs);
    std::vector<Fr> Mz[3];
    for (int k = 0; k < 3; k++) {
        Mz[k].assign(num_cons, fr_zero());
        const SparseMat &m = M[k];
        for (size_t i = 0; i < m.val.size(); i++) Mz[k][m.row[i]] = fr_add(Mz[k][m.row[i]], fr_mul(m.val[i], z[m.col[i]]));
    }
    for (size_t r = 0; r < num_cons; r++) if (!fr_eq(fr_mul(Mz[0][r], Mz[1][r]), Mz[2][r])) return false;
    return true;
}

// ================================================================================================ generators (commitments.rs)
static const uint8_t kBasepointCompressed[32] = {
    0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
    0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};

std::vector<Pt> derive_generators(const char *label, size_t count) {
    Shake256 xof; xof.absorb(label, strlen(label)); xof.absorb(kBasepointCompressed, 32);
    std::vector<uint8_t> u(64 * count);
    xof.squeeze(u.data(), u.size());                                 // the stream is sequential; the one-way maps (two Elligator maps + an addition each) are not
    std::vector<Pt> out(count);
    const unsigned nt = count < 64 ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto work = [&](unsigned t) { for (size_t i = count * t / nt; i < count * (t + 1) / nt; i++) out[i] = pt_from_uniform_bytes(&u[64 * i]); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    return out;
}

std::unique_ptr<Gens> gens_new(size_t num_cons, size_t num_vars, size_t num_inputs) {
    (void)num_cons;
    auto g = std::make_unique<Gens>();
    size_t nvp = next_pow2(std::max(num_vars, num_inputs + 1));
    size_t ell = ilog2(nvp);
    size_t R = (size_t)1 << (ell - ell / 2);
    g->num_vars_padded = nvp; g->R = R;
    g->P = derive_generators("gens_r1cs_sat", std::max(R + 2, (size_t)5));
    g->pc_n.G.resize(R); std::iota(g->pc_n.G.begin(), g->pc_n.G.end(), 0u); g->pc_n.h = (uint32_t)(R + 1);
    g->pc_1.G = {(uint32_t)R}; g->pc_1.h = (uint32_t)(R + 1);
    g->sc_1 = g->pc_1;
    g->sc_3.G = {0, 1, 2}; g->sc_3.h = 3;
    g->sc_4.G = {0, 1, 2, 3}; g->sc_4.h = 4;
    g->small_slot.assign(g->P.size(), -1);
    const size_t small[7] = {0, 1, 2, 3, 4, R, R + 1};
    std::vector<size_t> uniq;
    for (size_t idx : small) {
        if (g->small_slot[idx] >= 0) continue;
        g->small_slot[idx] = (int)g->small_tables.size();
        g->small_tables.emplace_back(); uniq.push_back(idx);
    }
    {   // the host window tables of these (up to seven) points are independent builds
        std::vector<std::thread> th;
        for (size_t idx : uniq) { Gens *gp = g.get(); th.emplace_back([gp, idx] { gp->small_tables[gp->small_slot[idx]].build(gp->P[idx]); }); }
        for (auto &x : th) x.join();
    }
    return g;
}

PtFe Gens::commit_terms_fe(const Term *t, size_t n) const {
    PtFe acc = ptfe_identity();
    for (size_t i = 0; i < n; i++) {
        int slot = small_slot[t[i].base];
        if (slot < 0) throw Error(OTTI_ERR_INTERNAL, "commit_terms: base without a host table");
        small_tables[slot].accumulate(acc, t[i].s);
    }
    return acc;
}
Pt Gens::commit_terms(const Term *t, size_t n) const { return ptfe_to(commit_terms_fe(t, n)); }
Pt Gens::commit_generic(const Fr *v, size_t n, const Fr &blind, const GensView &gv) const {
    std::vector<Fr> s(v, v + n); s.push_back(blind);
    std::vector<Pt> pts(n + 1);
    for (size_t i = 0; i < n; i++) pts[i] = P[gv.G[i]];
    pts[n] = P[gv.h];
    return host_msm(s.data(), pts.data(), n + 1);
}

// ================================================================================================ unipoly.rs
void unipoly_from_evals(Fr *c, const Fr *e, size_t n) {
    static const Fr two_inv = fr_inv(fr_from_u64(2)), six_inv = fr_inv(fr_from_u64(6));
    if (n == 3) {
        Fr a = fr_mul(two_inv, fr_add(fr_sub(fr_sub(e[2], e[1]), e[1]), e[0]));
        c[0] = e[0]; c[1] = fr_sub(fr_sub(e[1], e[0]), a); c[2] = a;
    } else {
        Fr e1x3 = fr_add(fr_dbl(e[1]), e[1]), e2x3 = fr_add(fr_dbl(e[2]), e[2]);
        Fr a = fr_mul(six_inv, fr_sub(fr_add(fr_sub(e[3], e2x3), e1x3), e[0]));
        Fr e1x5 = fr_add(fr_dbl(fr_dbl(e[1])), e[1]), e2x4 = fr_dbl(fr_dbl(e[2]));
        Fr b = fr_mul(two_inv, fr_sub(fr_add(fr_sub(fr_dbl(e[0]), e1x5), e2x4), e[3]));
        c[0] = e[0]; c[1] = fr_sub(fr_sub(fr_sub(e[1], e[0]), a), b); c[2] = b; c[3] = a;
    }
}
Fr unipoly_eval(const Fr *c, size_t n, const Fr &r) {
    Fr ev = c[0], pw = r;
    for (size_t i = 1; i < n; i++) { ev = fr_add(ev, fr_mul(pw, c[i])); pw = fr_mul(pw, r); }
    return ev;
}

// ================================================================================================ sigma protocols (nizk/mod.rs), prover side
static inline CPoint commit2(const Gens &g, uint32_t b0, const Fr &s0, uint32_t b1, const Fr &s1) {
    Term t[2] = {{b0, s0}, {b1, s1}}; CPoint c; g.commit_terms_c(c.b, t, 2); return c;
}
static inline CPoint commit_scalar(const Gens &g, const GensView &g1, const Fr &x, const Fr &blind) { return commit2(g, g1.G[0], x, g1.h, blind); }
static CPoint commit_vec(const Gens &g, const GensView &gn, const Fr *v, size_t n, const Fr &blind) {
    Term t[8]; for (size_t i = 0; i < n; i++) t[i] = {gn.G[i], v[i]}; t[n] = {gn.h, blind};
    CPoint c; g.commit_terms_c(c.b, t, n + 1); return c;
}

KnowledgeProof knowledge_prove(CPoint &C, const Gens &g, Transcript &tr, RandomTape &tape, const Fr &x, const Fr &r) {
    tr.append_protocol_name("knowledge proof");
    Fr t1 = tape.random_scalar("t1"), t2 = tape.random_scalar("t2");
    C = commit_scalar(g, g.sc_1, x, r); tr.append_point("C", C.b);
    KnowledgeProof pf;
    pf.alpha = commit_scalar(g, g.sc_1, t1, t2); tr.append_point("alpha", pf.alpha.b);
    Fr c = tr.challenge_scalar("c");
    pf.z1 = fr_add(fr_mul(x, c), t1); pf.z2 = fr_add(fr_mul(r, c), t2);
    return pf;
}

EqualityProof equality_prove(const Gens &g, Transcript &tr, RandomTape &tape, const Fr &v1, const Fr &s1, const Fr &v2, const Fr &s2) {
    tr.append_protocol_name("equality proof");
    Fr r = tape.random_scalar("r");
    CPoint C1 = commit_scalar(g, g.sc_1, v1, s1); tr.append_point("C1", C1.b);
    CPoint C2 = commit_scalar(g, g.sc_1, v2, s2); tr.append_point("C2", C2.b);
    EqualityProof pf;
    Term t = {g.sc_1.h, r}; g.commit_terms_c(pf.alpha.b, &t, 1); tr.append_point("alpha", pf.alpha.b);
    Fr c = tr.challenge_scalar("c");
    pf.z = fr_add(fr_mul(c, fr_sub(s1, s2)), r);
    return pf;
}

ProductProof product_prove(CPoint &X, CPoint &Y, CPoint &Z, const Gens &g, Transcript &tr, RandomTape &tape, const Fr &x, const Fr &rX,
                           const Fr &y, const Fr &rY, const Fr &z, const Fr &rZ) {
    tr.append_protocol_name("product proof");
    Fr b1 = tape.random_scalar("b1"), b2 = tape.random_scalar("b2"), b3 = tape.random_scalar("b3"), b4 = tape.random_scalar("b4"),
       b5 = tape.random_scalar("b5");
    const GensView &g1 = g.sc_1;
    X = commit_scalar(g, g1, x, rX); tr.append_point("X", X.b);
    Y = commit_scalar(g, g1, y, rY); tr.append_point("Y", Y.b);
    Z = commit_scalar(g, g1, z, rZ); tr.append_point("Z", Z.b);
    ProductProof pf;
    pf.alpha = commit_scalar(g, g1, b1, b2); tr.append_point("alpha", pf.alpha.b);
    pf.beta = commit_scalar(g, g1, b3, b4); tr.append_point("beta", pf.beta.b);
    // delta = b3*X + b5*h with X = x*G + rX*h, i.e. (b3*x)*G + (b3*rX + b5)*h: stays on the fixed-base tables
    pf.delta = commit_scalar(g, g1, fr_mul(b3, x), fr_add(fr_mul(b3, rX), b5)); tr.append_point("delta", pf.delta.b);
    Fr c = tr.challenge_scalar("c");
    pf.z[0] = fr_add(b1, fr_mul(c, x)); pf.z[1] = fr_add(b2, fr_mul(c, rX)); pf.z[2] = fr_add(b3, fr_mul(c, y));
    pf.z[3] = fr_add(b4, fr_mul(c, rY)); pf.z[4] = fr_add(b5, fr_mul(c, fr_sub(rZ, fr_mul(rX, y))));
    return pf;
}

// sumcheck.rs + nizk/mod.rs DotProductProof::prove.  NOTE on tape order: upstream draws blinds_poly and blinds_evals when the
// sum-check starts and (d_vec, r_delta, r_beta) inside each round's DotProductProof::prove; nothing else touches the tape in
// between, so drawing all rounds' values right after the blinds yields the same values.
void sumcheck_draw_tape(SumcheckState &st, ScalarSource &tape, size_t num_rounds, size_t ne) {
    st.blinds_poly = tape.random_vector("blinds_poly", num_rounds);
    st.blinds_evals = tape.random_vector("blinds_evals", num_rounds);
    st.pre.resize(num_rounds);
    for (auto &p : st.pre) {
        std::vector<Fr> d = tape.random_vector("d_vec", ne);
        for (size_t i = 0; i < ne; i++) p.d[i] = d[i];
        p.r_delta = tape.random_scalar("r_delta"); p.r_beta = tape.random_scalar("r_beta");
    }
}
static CPoint encode_sum(PtFe a, const PtFe &b) { CPoint c; ptfe_add(a, b); pt_encode_fe(c.b, a); return c; }

// the transcript work of one round, split at the challenge so the device can fold while the host finishes
RoundPart1 sumcheck_round_begin(ZKSumcheckProof &pf, size_t j, const Fr *evals, size_t ne, const SumcheckState &st, const Gens &g,
                                const GensView &gn, Transcript &tr) {
    RoundPart1 p; p.ne = ne;
    unipoly_from_evals(p.poly, evals, ne);
    // commit(poly, blinds_poly[j]) over gens_n: one fixed-base term per coefficient, spread over the helper threads
    PtFe part[4]; std::function<void()> tasks[4];
    for (size_t i = 0; i < ne; i++) tasks[i] = [&, i] { Term t = {gn.G[i], p.poly[i]}; part[i] = g.commit_terms_fe(&t, 1); };
    SpinPool::get().parallel(tasks, (int)ne);
    PtFe sum = st.pre[j].bp_fe;
    for (size_t i = 0; i < ne; i++) ptfe_add(sum, part[i]);
    pt_encode_fe(pf.comm_polys[j].b, sum);
    tr.append_point("comm_poly", pf.comm_polys[j].b);
    p.r_j = tr.challenge_scalar("challenge_nextround");
    return p;
}
void sumcheck_round_finish(ZKSumcheckProof &pf, size_t j, const RoundPart1 &p1, SumcheckState &st, const Gens &g, const GensView &gn,
                           Transcript &tr) {
    const size_t ne = p1.ne; const RoundPre &pre = st.pre[j]; (void)gn;
    Fr eval = unipoly_eval(p1.poly, ne, p1.r_j);
    Term te = {g.sc_1.G[0], eval};
    CPoint comm_eval = encode_sum(g.commit_terms_fe(&te, 1), pre.be_fe);                       // commit(eval, blinds_evals[j]) over gens_1
    tr.append_point("comm_claim_per_round", st.comm_claim.b);
    tr.append_point("comm_eval", comm_eval.b);
    std::vector<Fr> w = tr.challenge_vector("combine_two_claims_to_one", 2);
    Fr target = fr_add(fr_mul(w[0], st.claim), fr_mul(w[1], eval));
    const Fr &blind_sc = j == 0 ? st.blind_claim : st.blinds_evals[j - 1];
    Fr blind = fr_add(fr_mul(w[0], blind_sc), fr_mul(w[1], st.blinds_evals[j]));
    Fr a[4], pw = fr_one(), two = fr_from_u64(2);
    for (size_t i = 0; i < ne; i++) { a[i] = fr_add(fr_mul(w[0], i == 0 ? two : fr_one()), fr_mul(w[1], pw)); pw = fr_mul(pw, p1.r_j); }
    Fr ad = fr_zero(); for (size_t i = 0; i < ne; i++) ad = fr_add(ad, fr_mul(a[i], pre.d[i]));
    // DotProductProof::prove(gens_1, gens_n, poly, blinds_poly[j], a, target, blind); Cx is this round's comm_poly.
    // Cy = commit(target, blind) and beta = commit(<a,d>, r_beta) are independent: three concurrent pieces.
    DotProductProof dp; PtFe cy_g, cy_h; CPoint Cy;
    std::function<void()> tasks[3] = {
        [&] { Term t = {g.sc_1.G[0], target}; cy_g = g.commit_terms_fe(&t, 1); },
        [&] { Term t = {g.sc_1.h, blind}; cy_h = g.commit_terms_fe(&t, 1); },
        [&] { Term t = {g.sc_1.G[0], ad}; dp.beta = encode_sum(g.commit_terms_fe(&t, 1), pre.rb_fe); }};
    SpinPool &pool = SpinPool::get();
    int nw = pool.workers();
    if (nw >= 2) {
        pool.submit(0, tasks[1]); pool.submit(1, tasks[2]);
        tasks[0](); pool.wait(0);
        Cy = encode_sum(cy_g, cy_h);                                                      // overlaps the helper's beta compression
        pool.wait(1);
    } else { pool.parallel(tasks, 3); Cy = encode_sum(cy_g, cy_h); }
    tr.append_protocol_name("dot product proof");
    tr.append_point("Cx", pf.comm_polys[j].b);
    tr.append_point("Cy", Cy.b);
    tr.append_scalars("a", a, ne);
    dp.delta = pre.delta_c; tr.append_point("delta", dp.delta.b);
    tr.append_point("beta", dp.beta.b);
    Fr c = tr.challenge_scalar("c");
    dp.z.resize(ne); for (size_t i = 0; i < ne; i++) dp.z[i] = fr_add(fr_mul(c, p1.poly[i]), pre.d[i]);
    dp.z_delta = fr_add(fr_mul(c, st.blinds_poly[j]), pre.r_delta); dp.z_beta = fr_add(fr_mul(c, blind), pre.r_beta);
    pf.proofs[j] = dp;
    st.claim = eval; st.comm_claim = comm_eval; pf.comm_evals[j] = comm_eval;
}

// ================================================================================================ bincode layout
namespace {
struct Writer {
    std::vector<uint8_t> b;
    void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
    void u64(uint64_t x) { uint8_t t[8]; for (int i = 0; i < 8; i++) { t[i] = (uint8_t)x; x >>= 8; } raw(t, 8); }
    void pt(const CPoint &c) { raw(c.b, 32); }
    void fr(const Fr &x) { raw(x.v, 32); }                      // upstream serialises Scalar's Montgomery limbs
    void pts(const std::vector<CPoint> &v) { u64(v.size()); for (auto &c : v) pt(c); }
    void frs(const std::vector<Fr> &v) { u64(v.size()); for (auto &x : v) fr(x); }
    void sc(const ZKSumcheckProof &s) {
        pts(s.comm_polys); pts(s.comm_evals); u64(s.proofs.size());
        for (auto &d : s.proofs) { pt(d.delta); pt(d.beta); frs(d.z); fr(d.z_delta); fr(d.z_beta); }
    }
};
struct Reader {
    const uint8_t *p; size_t n, pos = 0;
    void need(size_t k) { if (pos + k > n) throw Error(OTTI_ERR_MALFORMED_PROOF, "proof truncated"); }
    uint64_t u64() { need(8); uint64_t x = 0; for (int i = 7; i >= 0; i--) x = (x << 8) | p[pos + i]; pos += 8; return x; }
    CPoint pt() { need(32); CPoint c; memcpy(c.b, p + pos, 32); pos += 32; return c; }
    Fr fr() { need(32); Fr x; memcpy(x.v, p + pos, 32); pos += 32; if (!fr_raw_is_canonical(x.v)) throw Error(OTTI_ERR_MALFORMED_PROOF, "scalar out of range"); return x; }
    size_t len(size_t max) { uint64_t k = u64(); if (k > max) throw Error(OTTI_ERR_MALFORMED_PROOF, "vector length out of range"); return (size_t)k; }
    std::vector<CPoint> pts(size_t max) { size_t k = len(max); std::vector<CPoint> v(k); for (auto &c : v) c = pt(); return v; }
    std::vector<Fr> frs(size_t max) { size_t k = len(max); std::vector<Fr> v(k); for (auto &x : v) x = fr(); return v; }
    ZKSumcheckProof sc() {
        ZKSumcheckProof s; s.comm_polys = pts(64); s.comm_evals = pts(64); size_t k = len(64); s.proofs.resize(k);
        for (auto &d : s.proofs) { d.delta = pt(); d.beta = pt(); d.z = frs(4); d.z_delta = fr(); d.z_beta = fr(); }
        return s;
    }
};
}  // namespace

std::vector<uint8_t> NizkProof::serialize() const {
    Writer w;
    w.pts(comm_vars); w.sc(sc1);
    for (int i = 0; i < 4; i++) w.pt(claims_phase2[i]);
    w.pt(pok.alpha); w.fr(pok.z1); w.fr(pok.z2);
    w.pt(prod.alpha); w.pt(prod.beta); w.pt(prod.delta); for (int i = 0; i < 5; i++) w.fr(prod.z[i]);
    w.pt(eq1.alpha); w.fr(eq1.z);
    w.sc(sc2);
    w.pt(comm_vars_at_ry);
    w.pts(polyeval.L_vec); w.pts(polyeval.R_vec); w.pt(polyeval.delta); w.pt(polyeval.beta); w.fr(polyeval.z1); w.fr(polyeval.z2);
    w.pt(eq2.alpha); w.fr(eq2.z);
    w.frs(rx); w.frs(ry);
    return std::move(w.b);
}

NizkProof NizkProof::parse(const uint8_t *p, size_t n) {
    Reader r{p, n}; NizkProof P;
    P.comm_vars = r.pts((size_t)1 << 20); P.sc1 = r.sc();
    for (int i = 0; i < 4; i++) P.claims_phase2[i] = r.pt();
    P.pok.alpha = r.pt(); P.pok.z1 = r.fr(); P.pok.z2 = r.fr();
    P.prod.alpha = r.pt(); P.prod.beta = r.pt(); P.prod.delta = r.pt(); for (int i = 0; i < 5; i++) P.prod.z[i] = r.fr();
    P.eq1.alpha = r.pt(); P.eq1.z = r.fr();
    P.sc2 = r.sc();
    P.comm_vars_at_ry = r.pt();
    P.polyeval.L_vec = r.pts(64); P.polyeval.R_vec = r.pts(64); P.polyeval.delta = r.pt(); P.polyeval.beta = r.pt();
    P.polyeval.z1 = r.fr(); P.polyeval.z2 = r.fr();
    P.eq2.alpha = r.pt(); P.eq2.z = r.fr();
    P.rx = r.frs(64); P.ry = r.frs(64);
    if (r.pos != n) throw Error(OTTI_ERR_MALFORMED_PROOF, "trailing bytes after proof");
    return P;
}

// ================================================================================================ verifier
// (external linkage: snark_host.cpp builds SNARK::verify from the same pieces; declared in snark.h)
Pt dec(const CPoint &c) { Pt p; if (!pt_decode_fast(p, c.b)) throw VerifyFail{OTTI_ERR_VERIFY_DECOMPRESS}; return p; }
void require(bool ok) { if (!ok) throw VerifyFail{OTTI_ERR_VERIFY_INTERNAL}; }
Pt commit_scalar_pt(const Gens &g, const GensView &g1, const Fr &x, const Fr &blind) { return g.commit_generic(&x, 1, blind, g1); }
// Group equations that do not feed the transcript (most of the verifier's work: two scalar multiplications and a small MSM per
// sum-check round) are collected here and checked at the end, spread over the host cores.  Each entry throws VerifyFail on failure.
// a sum over many points, split over the host cores (the verifier's two sqrt(V)-sized multi-scalar multiplications)
Pt host_msm_wide(const Fr *sc, const Pt *pts, size_t n) {
    const size_t nt = std::min<size_t>({n / 128, (size_t)16, (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (nt < 2) return host_msm(sc, pts, n);
    std::vector<Pt> part(nt); std::vector<std::thread> th;
    auto work = [&](size_t t) { size_t lo = n * t / nt, hi = n * (t + 1) / nt; part[t] = host_msm(sc + lo, pts + lo, hi - lo); };
    for (size_t t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    Pt acc = part[0]; for (size_t t = 1; t < nt; t++) acc = pt_add(acc, part[t]);
    return acc;
}
RowSumBeginHook g_row_sum_begin_hook = nullptr;
RowSumFinishHook g_row_sum_finish_hook = nullptr;
RowSum::RowSum(const CPoint *C_, size_t n_) : C(C_), n(n_) { if (g_row_sum_begin_hook && g_row_sum_finish_hook && !getenv("OTTI_VERIFY_HOST")) job = g_row_sum_begin_hook(C, n); }
RowSum::~RowSum() { if (job) { Pt t; (void)g_row_sum_finish_hook(job, nullptr, t); } }
Pt RowSum::finish(const Fr *s) {
    static const bool trace = getenv("OTTI_TRACE") != nullptr; const auto t0 = std::chrono::steady_clock::now();
    struct Lap { bool on; std::chrono::steady_clock::time_point t0; size_t n; bool dev; ~Lap() { if (on) fprintf(stderr, "[otti]   row sum over %zu points (%s) %.3f ms\n", n, dev ? "device" : "host", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } } lap{trace, t0, n, job != nullptr};
    if (job) {
        Pt out; RowSumJob *j = job; job = nullptr;
        const int rc = g_row_sum_finish_hook(j, s, out);
        if (rc == 0) return out;
        lap.dev = false;
        if (rc == OTTI_ERR_VERIFY_DECOMPRESS) throw VerifyFail{rc};
    }
    std::vector<Pt> Cs(n);
    {   // decompression of the row commitments (an inverse square root each), striped over the host cores
        const size_t nt = std::min<size_t>({n / 64 + 1, (size_t)16, (size_t)std::max(1u, std::thread::hardware_concurrency())});
        std::vector<int> bad(nt, 0); std::vector<std::thread> th;
        auto work = [&](size_t t) { try { for (size_t i = t; i < n; i += nt) Cs[i] = dec(C[i]); } catch (const VerifyFail &f) { bad[t] = f.code; } };
        for (size_t t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (int b : bad) if (b) throw VerifyFail{b};
    }
    return host_msm_wide(s, Cs.data(), n);
}
void run_deferred(Deferred &d) { d.finish(); }

void knowledge_verify(const KnowledgeProof &pf, const Gens &g, Transcript &tr, const CPoint &C, Deferred &later) {
    tr.append_protocol_name("knowledge proof");
    tr.append_point("C", C.b); tr.append_point("alpha", pf.alpha.b);
    Fr c = tr.challenge_scalar("c");
    const KnowledgeProof *p = &pf; const Gens *gp = &g;
    later.push_back([=] { require(pt_eq(commit_scalar_pt(*gp, gp->sc_1, p->z1, p->z2), pt_add(host_scalarmul(dec(C), c), dec(p->alpha)))); });
}
void equality_verify(const EqualityProof &pf, const Gens &g, Transcript &tr, const CPoint &C1, const CPoint &C2, Deferred &later) {
    tr.append_protocol_name("equality proof");
    tr.append_point("C1", C1.b); tr.append_point("C2", C2.b); tr.append_point("alpha", pf.alpha.b);
    Fr c = tr.challenge_scalar("c");
    const EqualityProof *p = &pf; const Gens *gp = &g;
    later.push_back([=] { require(pt_eq(host_scalarmul(gp->P[gp->sc_1.h], p->z), pt_add(host_scalarmul(pt_sub(dec(C1), dec(C2)), c), dec(p->alpha)))); });
}
bool product_check(const CPoint &P, const Pt &X, const Fr &c, const Pt &G, const Pt &h, const Fr &z1, const Fr &z2) {
    Pt lhs = pt_add(dec(P), host_scalarmul(X, c));
    Pt rhs = pt_add(host_scalarmul(G, z1), host_scalarmul(h, z2));
    return pt_eq(lhs, rhs);
}
void product_verify(const ProductProof &pf, const Gens &g, Transcript &tr, const CPoint &X, const CPoint &Y, const CPoint &Z, Deferred &later) {
    tr.append_protocol_name("product proof");
    tr.append_point("X", X.b); tr.append_point("Y", Y.b); tr.append_point("Z", Z.b);
    tr.append_point("alpha", pf.alpha.b); tr.append_point("beta", pf.beta.b); tr.append_point("delta", pf.delta.b);
    Fr c = tr.challenge_scalar("c");
    const ProductProof *p = &pf; const Gens *gp = &g;
    later.push_back([=] { const Pt &G = gp->P[gp->sc_1.G[0]], &h = gp->P[gp->sc_1.h]; require(product_check(p->alpha, dec(X), c, G, h, p->z[0], p->z[1])); });
    later.push_back([=] { const Pt &G = gp->P[gp->sc_1.G[0]], &h = gp->P[gp->sc_1.h]; require(product_check(p->beta, dec(Y), c, G, h, p->z[2], p->z[3])); });
    later.push_back([=] { const Pt &h = gp->P[gp->sc_1.h]; require(product_check(p->delta, dec(Z), c, dec(X), h, p->z[2], p->z[4])); });
}
void dotproduct_verify(const DotProductProof &pf, const Gens &g, const GensView &gn, Transcript &tr, const Fr *a, size_t n,
                       const CPoint &Cx, const CPoint &Cy, Deferred &later) {
    require(pf.z.size() == n && gn.G.size() == n);
    tr.append_protocol_name("dot product proof");
    tr.append_point("Cx", Cx.b); tr.append_point("Cy", Cy.b);
    tr.append_scalars("a", a, n);
    tr.append_point("delta", pf.delta.b); tr.append_point("beta", pf.beta.b);
    Fr c = tr.challenge_scalar("c");
    Fr za = fr_zero(); for (size_t i = 0; i < n; i++) za = fr_add(za, fr_mul(pf.z[i], a[i]));
    const DotProductProof *p = &pf; const Gens *gp = &g; const GensView *gv = &gn;       // all outlive the deferred run (owned by nizk_verify's frame)
    later.push_back([=] { require(pt_eq(pt_add(host_scalarmul(dec(Cx), c), dec(p->delta)), gp->commit_generic(p->z.data(), n, p->z_delta, *gv))); });
    later.push_back([=] { require(pt_eq(pt_add(host_scalarmul(dec(Cy), c), dec(p->beta)), commit_scalar_pt(*gp, gp->sc_1, za, p->z_beta))); });
}
// The commitments a sum-check's rounds decompress ON the sequential path (comm_evals: each one is combined with a challenge and the
// result is hashed before the next round can start) are decompressed ahead of it by the background threads, one per task; the round
// takes the prepared multiples if they are there and starts from the encoding itself if not.  State: 0 pending, 1 ready, 2 not an encoding.
struct PreDecoded {
    const std::vector<CPoint> *src = nullptr; std::vector<SplitTable> tabs; std::unique_ptr<std::atomic<int>[]> state;
    void start(const std::vector<CPoint> &v, Deferred &later) {
        src = &v; tabs.resize(v.size()); state.reset(new std::atomic<int>[v.size()]);
        for (size_t i = 0; i < v.size(); i++) state[i].store(0, std::memory_order_relaxed);
        for (size_t i = 0; i < v.size(); i++)
            later.push_back([this, i] { Pt p; const bool ok = pt_decode_fast(p, (*src)[i].b); if (ok) split_table_build(tabs[i], p); state[i].store(ok ? 1 : 2, std::memory_order_release); });
    }
    // s * (point i): from the prepared table when a worker has finished it (hostgroup.h SplitTable: a third of the chain), else from scratch
    Pt mul(size_t i, const Fr &s) const {
        const int st = state ? state[i].load(std::memory_order_acquire) : 0;
        if (st == 1) return split_table_mul(tabs[i], s);
        if (st == 2) throw VerifyFail{OTTI_ERR_VERIFY_DECOMPRESS};
        return host_scalarmul(dec((*src)[i]), s);
    }
};
// ZKSumcheckInstanceProof::verify
CPoint sumcheck_verify(const ZKSumcheckProof &pf, const CPoint &comm_claim, size_t num_rounds, size_t degree, const Gens &g,
                       const GensView &gn, Transcript &tr, std::vector<Fr> &r, Deferred &later, const PreDecoded *pre = nullptr) {
    require(gn.G.size() == degree + 1 && pf.comm_polys.size() == num_rounds && pf.comm_evals.size() == num_rounds &&
            pf.proofs.size() == num_rounds);
    size_t ne = degree + 1; r.clear();
    for (size_t i = 0; i < num_rounds; i++) {
        tr.append_point("comm_poly", pf.comm_polys[i].b);
        Fr r_i = tr.challenge_scalar("challenge_nextround");
        const CPoint &ccpr = i == 0 ? comm_claim : pf.comm_evals[i - 1];
        tr.append_point("comm_claim_per_round", ccpr.b); tr.append_point("comm_eval", pf.comm_evals[i].b);
        std::vector<Fr> w = tr.challenge_vector("combine_two_claims_to_one", 2);
        // the combined claim's commitment is hashed ("Cy" below), so it stays on the sequential path: its two halves on two threads
        Pt half[2]; int bad[2] = {0, 0};
        std::function<void()> halves[2] = {
            [&] { try { half[0] = (pre && i) ? pre->mul(i - 1, w[0]) : host_scalarmul(dec(ccpr), w[0]); } catch (const VerifyFail &f) { bad[0] = f.code; } },
            [&] { try { half[1] = pre ? pre->mul(i, w[1]) : host_scalarmul(dec(pf.comm_evals[i]), w[1]); } catch (const VerifyFail &f) { bad[1] = f.code; } }};
        SpinPool::get().parallel(halves, 2);
        if (bad[0] || bad[1]) throw VerifyFail{bad[0] ? bad[0] : bad[1]};
        CPoint comm_target; pt_encode(comm_target.b, pt_add(half[0], half[1]));
        Fr a[4], pw = fr_one(), two = fr_from_u64(2);
        for (size_t k = 0; k < ne; k++) { a[k] = fr_add(fr_mul(w[0], k == 0 ? two : fr_one()), fr_mul(w[1], pw)); pw = fr_mul(pw, r_i); }
        dotproduct_verify(pf.proofs[i], g, gn, tr, a, ne, pf.comm_polys[i], comm_target, later);
        r.push_back(r_i);
    }
    return pf.comm_evals[num_rounds - 1];
}
// g_hat = <s, gens_n.G> is a FIXED-base sum over the first n generators of the stream: when the prover's window table of that stream is
// already resident (prove and verify in one process, as `spzk verify` runs them), the device side registers this hook and the verifier
// takes the sum from one small MSM launch instead of a host Pippenger over n points.  Without it (no device, no table) nothing changes.
FixedBaseMsmHook g_fixed_base_msm_hook = nullptr;
// BulletReductionProof::verify + the closing equation of DotProductProofLog::verify.  The transcript part (the round challenges, then c)
// and the fixed-base sum g_hat (device, when the prover's table is resident) run on the calling thread; Gamma_hat — a variable-base sum
// over the 2 log n proof points — and the closing group equation feed nothing that is hashed, so with `later` they become one more
// deferred check (spartan.h verifiers: SNARK mode has three of these proofs in a row).
void dotproductlog_verify(const DotProductProofLog &pf, size_t n, const Gens &g, const PcView &v, Transcript &tr, const Fr *a, const CPoint &Cx, const CPoint &Cy, Deferred *later) {
    require(v.R == n && g.P.size() >= n + 2);
    static const bool trace = getenv("OTTI_TRACE") != nullptr; auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (!trace) return; const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[otti]   dotproductlog_verify n=%zu %-22s %.3f ms\n", n, what, std::chrono::duration<double, std::milli>(t - t_lap).count()); t_lap = t; };
    tr.append_protocol_name("dot product proof (log)");
    tr.append_point("Cx", Cx.b); tr.append_point("Cy", Cy.b);
    const size_t lg = pf.L_vec.size();
    require(pf.R_vec.size() == lg && lg < 32 && n == ((size_t)1 << lg));
    std::vector<Fr> ch(lg), chi(lg);
    for (size_t i = 0; i < lg; i++) { tr.append_point("L", pf.L_vec[i].b); tr.append_point("R", pf.R_vec[i].b); ch[i] = tr.challenge_scalar("u"); }
    tr.append_point("delta", pf.delta.b); tr.append_point("beta", pf.beta.b);
    const Fr c = tr.challenge_scalar("c");
    // everything below is arithmetic on values the transcript has already absorbed
    Fr allinv = fr_one();
    {   // the lg inverses from ONE inversion (Montgomery's trick): allinv = 1 / prod ch[i], chi[i] = allinv * prod_{k != i} ch[k]
        std::vector<Fr> pre(lg + 1); pre[0] = fr_one();
        for (size_t i = 0; i < lg; i++) pre[i + 1] = fr_mul(pre[i], ch[i]);
        allinv = fr_inv(pre[lg]);
        Fr suf = allinv;                                                  // 1 / (ch[0] .. ch[i]) walking down
        for (size_t i = lg; i-- > 0;) { chi[i] = fr_mul(suf, pre[i]); suf = fr_mul(suf, ch[i]); }
    }
    for (size_t i = 0; i < lg; i++) { ch[i] = fr_sqr(ch[i]); chi[i] = fr_sqr(chi[i]); }
    std::vector<Fr> s(n); s[0] = allinv;
    for (size_t i = 1; i < n; i++) { size_t lg_i = ilog2(i + 1) - 1; size_t k = (size_t)1 << lg_i; s[i] = fr_mul(s[i - k], ch[(lg - 1) - lg_i]); }
    Pt g_hat;
    if (!(g_fixed_base_msm_hook && g_fixed_base_msm_hook(g, s.data(), n, g_hat))) g_hat = host_msm_wide(s.data(), g.P.data(), n);
    Fr a_hat = fr_zero(); for (size_t i = 0; i < n; i++) a_hat = fr_add(a_hat, fr_mul(a[i], s[i]));
    lap("challenges, s, g_hat");
    const DotProductProofLog *p = &pf; const Gens *gp = &g;
    // the closing equation  ((Gamma_hat c + beta) a_hat + delta == (g_hat + a_hat G) z1 + z2 h)  as FOUR deferred jobs: the two halves of
    // Gamma_hat (log n decompressions and a small variable-base sum each) and the right-hand side run side by side on the workers, the
    // fourth job waits for them (jobs are claimed in order, so the three are under way or done when it starts) and finishes the chain.
    struct Closing { Pt half[2], rhs; std::atomic<int> done{0}, ok{0}; };
    struct Arrive { Closing &c; bool good = false; ~Arrive() { if (good) c.ok.fetch_add(1, std::memory_order_release); c.done.fetch_add(1, std::memory_order_release); } };
    auto st = std::make_shared<Closing>();
    auto half = [=](int which) {
        Arrive arr{*st};
        std::vector<Pt> pts(lg + 1); std::vector<Fr> sc(which ? chi : ch);
        for (size_t i = 0; i < lg; i++) pts[i] = dec(which ? p->R_vec[i] : p->L_vec[i]);
        if (which == 0) { sc.push_back(fr_one()); pts[lg] = pt_add(dec(Cx), dec(Cy)); }
        st->half[which] = host_msm(sc.data(), pts.data(), sc.size());
        arr.good = true;
    };
    auto right = [=] {
        Arrive arr{*st};
        st->rhs = pt_add(host_scalarmul(pt_add(g_hat, host_scalarmul(gp->P[v.g1], a_hat)), p->z1), host_scalarmul(gp->P[v.h1], p->z2));
        arr.good = true;
    };
    auto closing = [=] {
        while (st->done.load(std::memory_order_acquire) < 3) {
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
        if (st->ok.load(std::memory_order_acquire) < 3) return;            // the job that failed has reported why
        const Pt Gamma_hat = pt_add(st->half[0], st->half[1]);
        const Pt lhs = pt_add(host_scalarmul(pt_add(host_scalarmul(Gamma_hat, c), dec(p->beta)), a_hat), dec(p->delta));
        require(pt_eq(lhs, st->rhs));
    };
    if (later) { later->push_back([=] { half(0); }); later->push_back([=] { half(1); }); later->push_back(right); later->push_back(closing); }
    else { half(0); half(1); right(); closing(); }
    lap(later ? "closing equation (deferred)" : "Gamma_hat, closing equation");
}
// R1CSProof::verify: `tr` already carries the caller's protocol name (NIZK / SNARK); returns the challenges the transcript produced
int r1cs_verify_host(const NizkProof &P, size_t N, size_t V, const std::vector<Fr> &inputs, const Fr inst_evals_given[3], const Gens &g, Transcript &tr,
                     std::vector<Fr> &rx, std::vector<Fr> &ry, const InstEvalFetch *fetch) {
    try {
        const size_t nrx = ilog2(N), nry = ilog2(2 * V);
        size_t ell = ilog2(V), Lsz = (size_t)1 << (ell / 2), Rsz = (size_t)1 << (ell - ell / 2);
        require(g.num_vars_padded == V && P.comm_vars.size() == Lsz);
        const Fr one = fr_one();
        // R1CSProof::verify
        tr.append_protocol_name("R1CS proof");
        tr.append_message("poly_commitment", "poly_commitment_begin", 21);
        for (auto &c : P.comm_vars) tr.append_point("poly_commitment_share", c.b);
        tr.append_message("poly_commitment", "poly_commitment_end", 19);
        std::vector<Fr> tau = tr.challenge_vector("challenge_tau", nrx);
        CPoint claim_phase1; pt_encode(claim_phase1.b, commit_scalar_pt(g, g.sc_1, fr_zero(), fr_zero()));
        RowSum rows(P.comm_vars.data(), Lsz);                             // with a device: the row commitments start decompressing now
        PreDecoded pre1, pre2;                                            // declared before `later`: its workers fill them, so they must be destroyed after it
        Deferred later;                                                   // P, g and the CPoints it captures live until run_deferred below
        SpinPool::Session pool_session;
        const bool trace = getenv("OTTI_TRACE") != nullptr; auto t_lap = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) { if (!trace) return; const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[otti] r1cs_verify %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_lap).count()); t_lap = t; };
        if (P.sc1.comm_evals.size() == nrx && P.sc2.comm_evals.size() == nry) { pre1.start(P.sc1.comm_evals, later); pre2.start(P.sc2.comm_evals, later); }
        CPoint comm_post1 = sumcheck_verify(P.sc1, claim_phase1, nrx, 3, g, g.sc_4, tr, rx, later, pre1.src ? &pre1 : nullptr);
        lap("sum-check one");
        const CPoint &cAz = P.claims_phase2[0], &cBz = P.claims_phase2[1], &cCz = P.claims_phase2[2], &cPr = P.claims_phase2[3];
        knowledge_verify(P.pok, g, tr, cCz, later);
        product_verify(P.prod, g, tr, cAz, cBz, cPr, later);
        tr.append_point("comm_Az_claim", cAz.b); tr.append_point("comm_Bz_claim", cBz.b);
        tr.append_point("comm_Cz_claim", cCz.b); tr.append_point("comm_prod_Az_Bz_claims", cPr.b);
        Fr taus_bound = fr_one();
        for (size_t i = 0; i < nrx; i++) taus_bound = fr_mul(taus_bound, fr_add(fr_mul(rx[i], tau[i]), fr_mul(fr_sub(one, rx[i]), fr_sub(one, tau[i]))));
        CPoint expected1; pt_encode(expected1.b, host_scalarmul(pt_sub(dec(cPr), dec(cCz)), taus_bound));
        equality_verify(P.eq1, g, tr, expected1, comm_post1, later);
        Fr rA = tr.challenge_scalar("challenege_Az"), rB = tr.challenge_scalar("challenege_Bz"), rC = tr.challenge_scalar("challenege_Cz");
        CPoint comm_claim2;
        pt_encode(comm_claim2.b, pt_add(pt_add(host_scalarmul(dec(cAz), rA), host_scalarmul(dec(cBz), rB)), host_scalarmul(dec(cCz), rC)));
        lap("sigma protocols");
        CPoint comm_post2 = sumcheck_verify(P.sc2, comm_claim2, nry, 2, g, g.sc_3, tr, ry, later, pre2.src ? &pre2 : nullptr);
        lap("sum-check two");
        // PolyEvalProof::verify
        {
            tr.append_protocol_name("polynomial evaluation proof");
            size_t rl = nry - 1, lv = rl / 2;
            std::vector<Fr> Lv = eq_evals_host(ry.data() + 1, lv), Rv = eq_evals_host(ry.data() + 1 + lv, rl - lv);
            CPoint C_LZ; pt_encode(C_LZ.b, rows.finish(Lv.data()));
            lap("C_LZ");
            require(P.polyeval.L_vec.size() == ilog2(Rsz));
            const PcView pv = {g.pc_n.h, g.pc_1.G[0], g.pc_1.h, Rsz};
            dotproductlog_verify(P.polyeval, Rsz, g, pv, tr, Rv.data(), C_LZ, P.comm_vars_at_ry, &later);
        }
        // SparsePolynomial over (1, inputs) evaluated at ry[1..], MSB-first index bits
        Fr poly_input_eval = fr_zero();
        for (size_t idx = 0; idx <= inputs.size(); idx++) {
            Fr chi = fr_one();
            for (size_t j = 0; j < ell; j++) { bool bit = (idx >> (ell - j - 1)) & 1; chi = fr_mul(chi, bit ? ry[1 + j] : fr_sub(one, ry[1 + j])); }
            poly_input_eval = fr_add(poly_input_eval, fr_mul(chi, idx == 0 ? one : inputs[idx - 1]));
        }
        Pt comm_eval_Z = pt_add(host_scalarmul(dec(P.comm_vars_at_ry), fr_sub(one, ry[0])),
                                host_scalarmul(commit_scalar_pt(g, g.pc_1, poly_input_eval, fr_zero()), ry[0]));
        Fr inst_evals[3];                                                  // A, B, C at (rx, ry): given, or collected now from an evaluation begun before the rounds
        if (fetch) (*fetch)(inst_evals); else for (int k = 0; k < 3; k++) inst_evals[k] = inst_evals_given[k];
        Fr comb = fr_add(fr_add(fr_mul(rA, inst_evals[0]), fr_mul(rB, inst_evals[1])), fr_mul(rC, inst_evals[2]));
        CPoint expected2; pt_encode(expected2.b, host_scalarmul(comm_eval_Z, comb));
        equality_verify(P.eq2, g, tr, expected2, comm_post2, later);
        lap("log dot-product proof + rest");
        run_deferred(later);
        lap("deferred group equations");
        return OTTI_OK;
    } catch (const VerifyFail &f) { return f.code; }
    catch (const Error &e) { return e.code; }
}

int nizk_verify(const Instance &I, const std::vector<Fr> &inputs, const Gens &g, const void *tlabel, size_t tlabel_len, const uint8_t *proof,
                size_t proof_len, const Fr *inst_evals_opt, const InstEvalFetch *fetch) {
    try {
        if (inputs.size() != I.num_inputs) return OTTI_ERR_INVALID_NUM_INPUTS;
        NizkProof P = NizkProof::parse(proof, proof_len);
        size_t N = I.num_cons, V = I.num_vars, nrx = ilog2(N), nry = ilog2(2 * V);
        if (P.rx.size() != nrx || P.ry.size() != nry) return OTTI_ERR_VERIFY_INTERNAL;
        Transcript tr(tlabel, tlabel_len);
        tr.append_protocol_name("Spartan NIZK proof");
        Fr inst_evals[3];
        if (inst_evals_opt) { inst_evals[0] = inst_evals_opt[0]; inst_evals[1] = inst_evals_opt[1]; inst_evals[2] = inst_evals_opt[2]; }
        else if (!fetch) I.evaluate(P.rx, P.ry, inst_evals);
        std::vector<Fr> rx, ry;
        if (int rc = r1cs_verify_host(P, N, V, inputs, inst_evals, g, tr, rx, ry, inst_evals_opt ? nullptr : fetch)) return rc;
        for (size_t i = 0; i < nrx; i++) if (!fr_eq(rx[i], P.rx[i])) return OTTI_ERR_VERIFY_INTERNAL;
        for (size_t i = 0; i < nry; i++) if (!fr_eq(ry[i], P.ry[i])) return OTTI_ERR_VERIFY_INTERNAL;
        return OTTI_OK;
    } catch (const Error &e) { return e.code; }
}

// ================================================================================================ synthetic R1CS
// f(i) for i in [0, n) on the host's cores (every index independent: the result does not depend on the split); large instances only
template <class F> static void host_par_for(size_t n, F &&f) {
    unsigned nt = n < ((size_t)1 << 16) ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    auto run = [&](unsigned t) { for (size_t i = n * t / nt, e = n * (t + 1) / nt; i < e; i++) f(i); };
    unsigned started = 1;
    for (; started < nt; started++) { try { th.emplace_back(run, started); } catch (const std::system_error &) { break; } }
    run(0);
    for (unsigned t = started; t < nt; t++) run(t);
    for (auto &x : th) x.join();
}

void synth_r1cs(size_t n, size_t ni, uint64_t seed, std::vector<otti_entry> &A, std::vector<otti_entry> &B, std::vector<otti_entry> &C,
                std::vector<uint8_t> &vars32, std::vector<uint8_t> &inputs32) {
    size_t size_z = n + ni + 1;
    std::vector<Fr> Z(size_z);
    host_par_for(size_z, [&](size_t k) {
        Shake256 xof; uint8_t le[16], w[64];
        for (int i = 0; i < 8; i++) { le[i] = (uint8_t)(seed >> (8 * i)); le[8 + i] = (uint8_t)((uint64_t)k >> (8 * i)); }
        xof.absorb("otti-synth", 10); xof.absorb(le, 16); xof.squeeze(w, 64);
        Z[k] = fr_from_bytes_wide(w);
    });
    Z[n] = fr_one();
    A.resize(n); B.resize(n); C.resize(n);
    uint8_t one_bytes[32]; fr_to_bytes(one_bytes, fr_one());
    // batch the inversions Z[c]^-1
    std::vector<Fr> zc(n), pre(n);
    Fr acc = fr_one();
    for (size_t i = 0; i < n; i++) { zc[i] = Z[(i + 3) % size_z]; pre[i] = acc; if (!fr_is_zero(zc[i])) acc = fr_mul(acc, zc[i]); }
    acc = fr_inv(acc);
    std::vector<Fr> inv(n);
    for (size_t i = n; i-- > 0;) { if (fr_is_zero(zc[i])) { inv[i] = fr_zero(); continue; } inv[i] = fr_mul(acc, pre[i]); acc = fr_mul(acc, zc[i]); }
    host_par_for(n, [&](size_t i) {
        size_t a = i % size_z, b = (i + 2) % size_z, c = (i + 3) % size_z;
        A[i].row = i; A[i].col = a; memcpy(A[i].val, one_bytes, 32);
        B[i].row = i; B[i].col = b; memcpy(B[i].val, one_bytes, 32);
        Fr ab = fr_mul(Z[a], Z[b]);
        C[i].row = i;
        if (fr_is_zero(Z[c])) { C[i].col = n; fr_to_bytes(C[i].val, ab); }
        else { C[i].col = c; fr_to_bytes(C[i].val, fr_mul(ab, inv[i])); }
    });
    vars32.resize(32 * n); inputs32.resize(32 * ni);
    host_par_for(n, [&](size_t k) { fr_to_bytes(&vars32[32 * k], Z[k]); });
    for (size_t k = 0; k < ni; k++) fr_to_bytes(&inputs32[32 * k], Z[n + 1 + k]);
}

// "compiler-like" satisfiable instance (second distribution): 90 % of the witness below 2^64, 1..8 non-zeros per row with
// small signed coefficients, a heavily used constant column, one very long row.  Reported separately from the uniform instance.
void synth_r1cs_compiler_like(size_t n, size_t ni, uint64_t seed, std::vector<otti_entry> &A, std::vector<otti_entry> &B,
                              std::vector<otti_entry> &C, std::vector<uint8_t> &vars32, std::vector<uint8_t> &inputs32) {
    uint64_t st = seed * 0x9e3779b97f4a7c15ULL + 0x1234567;
    auto rnd = [&]() { st += 0x9e3779b97f4a7c15ULL; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
    const size_t size_z = n + 1 + ni;
    std::vector<Fr> Z(size_z), Zinv(size_z);
    for (size_t k = 0; k < size_z; k++) {
        if (rnd() % 10 != 0) Z[k] = fr_from_u64(rnd() | 1);
        else { uint8_t w[64]; for (int i = 0; i < 8; i++) { uint64_t x = rnd(); memcpy(w + 8 * i, &x, 8); } Z[k] = fr_from_bytes_wide(w); if (fr_is_zero(Z[k])) Z[k] = fr_one(); }
    }
    Z[n] = fr_one();
    {   // batch inversion of every z entry (all non-zero)
        std::vector<Fr> pre(size_z); Fr acc = fr_one();
        for (size_t k = 0; k < size_z; k++) { pre[k] = acc; acc = fr_mul(acc, Z[k]); }
        acc = fr_inv(acc);
        for (size_t k = size_z; k-- > 0;) { Zinv[k] = fr_mul(acc, pre[k]); acc = fr_mul(acc, Z[k]); }
    }
    A.clear(); B.clear(); C.clear();
    auto emit = [&](std::vector<otti_entry> &M, size_t row, size_t col, const Fr &v) { otti_entry e; e.row = row; e.col = col; fr_to_bytes(e.val, v); M.push_back(e); };
    auto lin = [&](std::vector<otti_entry> &M, size_t row, size_t kmax, bool with_const, std::vector<size_t> *used) {
        Fr sum = fr_zero(); size_t k = 1 + rnd() % kmax;
        for (size_t t = 0; t < k + (with_const ? 1 : 0); t++) {
            size_t col = (with_const && t == k) ? n : rnd() % size_z;
            uint64_t c = 1 + rnd() % 1000; Fr cv = fr_from_u64(c); if (rnd() % 5 == 0) cv = fr_neg(cv);
            emit(M, row, col, cv); sum = fr_add(sum, fr_mul(cv, Z[col]));
            if (used) used->push_back(col);
        }
        return sum;
    };
    for (size_t row = 0; row < n; row++) {
        Fr a = lin(A, row, row == 3 ? 300 : 8, rnd() % 2 == 0, nullptr);
        Fr b = lin(B, row, 8, rnd() % 2 == 0, nullptr);
        std::vector<size_t> used; Fr c = lin(C, row, 6, false, &used);
        size_t fix = row % n; while (std::find(used.begin(), used.end(), fix) != used.end()) fix = (fix + 1) % n;
        emit(C, row, fix, fr_mul(fr_sub(fr_mul(a, b), c), Zinv[fix]));
    }
    vars32.resize(32 * n); inputs32.resize(32 * ni);
    for (size_t k = 0; k < n; k++) fr_to_bytes(&vars32[32 * k], Z[k]);
    for (size_t k = 0; k < ni; k++) fr_to_bytes(&inputs32[32 * k], Z[n + 1 + k]);
}

}  // namespace otti
