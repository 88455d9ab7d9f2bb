// SNARK mode, host side: generators, the computation commitment's wire form, the proof's wire form, SNARK::verify.  See snark.h.
#include "snark.h"
#include <algorithm>
#include <thread>
#include <chrono>

namespace otti {

// ================================================================================================ lib.rs SNARKGens::new
static PcSet pc_set(size_t num_vars) {
    PcSet s; s.num_vars = num_vars; s.L = (size_t)1 << (num_vars / 2); s.R = (size_t)1 << (num_vars - num_vars / 2);   // EqPolynomial::compute_factored_lens
    s.h_n = (uint32_t)(s.R + 1); s.g1 = (uint32_t)s.R; s.h1 = (uint32_t)(s.R + 1);                                     // DotProductProofGens::new(R, label)
    return s;
}
std::unique_ptr<SnarkGens> snark_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs, size_t num_nz_entries) {
    auto g = std::make_unique<SnarkGens>();
    g->num_vars = next_pow2(std::max(num_vars, num_inputs + 1)); g->num_cons = num_cons < 2 ? 2 : next_pow2(num_cons); g->num_inputs = num_inputs;
    g->sat = gens_new(num_cons, num_vars, num_inputs);
    // R1CSCommitmentGens::new -> SparseMatPolyCommitmentGens::new(label, x, y, nz, batch_size = 3); at least two operations per matrix
    const size_t vx = ilog2(g->num_cons), vy = ilog2(2 * g->num_vars), lgnz = ilog2(next_pow2(std::max<size_t>(num_nz_entries, 2)));
    g->ops = pc_set(lgnz + ilog2(next_pow2(3 * 5))); g->mem = pc_set(std::max(vx, vy) + 1); g->derefs = pc_set(lgnz + ilog2(next_pow2(3 * 2)));
    const size_t Rmax = std::max({g->ops.R, g->mem.R, g->derefs.R});
    auto e = std::make_unique<Gens>();
    e->R = Rmax; e->num_vars_padded = 0;
    e->P = derive_generators("gens_r1cs_eval", Rmax + 2);             // the three generator sets are prefixes of one stream
    e->small_slot.assign(e->P.size(), -1);
    for (const PcSet *s : {&g->ops, &g->mem, &g->derefs})
        for (uint32_t idx : {s->g1, s->h1}) {
            if (e->small_slot[idx] >= 0) continue;
            e->small_slot[idx] = (int)e->small_tables.size();
            e->small_tables.emplace_back(); e->small_tables.back().build(e->P[idx]);
        }
    g->eval = std::move(e);
    return g;
}

// ================================================================================================ wire forms (bincode)
namespace {
struct Writer {
    std::vector<uint8_t> b;
    void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
    void u64(uint64_t x) { uint8_t t[8]; for (int i = 0; i < 8; i++) { t[i] = (uint8_t)x; x >>= 8; } raw(t, 8); }
    void pt(const CPoint &c) { raw(c.b, 32); }
    void fr(const Fr &x) { raw(x.v, 32); }                      // upstream serialises Scalar's Montgomery limbs
    void pts(const std::vector<CPoint> &v) { u64(v.size()); for (auto &c : v) pt(c); }
    void frs(const Fr *v, size_t n) { u64(n); for (size_t i = 0; i < n; i++) fr(v[i]); }
    void frs(const std::vector<Fr> &v) { frs(v.data(), v.size()); }
    void evals4(const Evals4 &e) { fr(e.init); frs(e.read, 3); frs(e.write, 3); fr(e.audit); }
    void batch(const ProductCircuitEvalProofBatched &p) {
        u64(p.layers.size());
        for (auto &L : p.layers) { u64(L.coeffs.size() / 3); for (size_t j = 0; j < L.coeffs.size() / 3; j++) frs(&L.coeffs[3 * j], 3); frs(L.left); frs(L.right); }
        frs(p.dotp_left); frs(p.dotp_right); frs(p.dotp_weight);
    }
    void dplog(const DotProductProofLog &d) { pts(d.L_vec); pts(d.R_vec); pt(d.delta); pt(d.beta); fr(d.z1); fr(d.z2); }
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
    void frs_fixed(Fr *v, size_t k) { if (u64() != k) throw Error(OTTI_ERR_MALFORMED_PROOF, "vector length"); for (size_t i = 0; i < k; i++) v[i] = fr(); }
    void evals4(Evals4 &e) { e.init = fr(); frs_fixed(e.read, 3); frs_fixed(e.write, 3); e.audit = fr(); }
    ProductCircuitEvalProofBatched batch() {
        ProductCircuitEvalProofBatched p; p.layers.resize(len(64));
        for (auto &L : p.layers) {
            size_t rounds = len(64); L.coeffs.resize(3 * rounds);
            for (size_t j = 0; j < rounds; j++) frs_fixed(&L.coeffs[3 * j], 3);
            L.left = frs(64); L.right = frs(64);
            if (L.left.size() != L.right.size()) throw Error(OTTI_ERR_MALFORMED_PROOF, "layer claims");
        }
        p.dotp_left = frs(64); p.dotp_right = frs(64); p.dotp_weight = frs(64);
        if (p.dotp_left.size() != p.dotp_right.size() || p.dotp_left.size() != p.dotp_weight.size()) throw Error(OTTI_ERR_MALFORMED_PROOF, "dot-product claims");
        return p;
    }
    DotProductProofLog dplog() {
        DotProductProofLog d; d.L_vec = pts(64); d.R_vec = pts(64); d.delta = pt(); d.beta = pt(); d.z1 = fr(); d.z2 = fr();
        if (d.L_vec.size() != d.R_vec.size()) throw Error(OTTI_ERR_MALFORMED_PROOF, "bullet reduction vectors");
        return d;
    }
    ZKSumcheckProof sc() {
        ZKSumcheckProof s; s.comm_polys = pts(64); s.comm_evals = pts(64); size_t k = len(64); s.proofs.resize(k);
        for (auto &d : s.proofs) { d.delta = pt(); d.beta = pt(); d.z = frs(4); d.z_delta = fr(); d.z_beta = fr(); }
        return s;
    }
};
void write_r1cs(Writer &w, const NizkProof &P) {                 // R1CSProof's fields, as NizkProof::serialize writes them (without rx, ry)
    w.pts(P.comm_vars); w.sc(P.sc1);
    for (int i = 0; i < 4; i++) w.pt(P.claims_phase2[i]);
    w.pt(P.pok.alpha); w.fr(P.pok.z1); w.fr(P.pok.z2);
    w.pt(P.prod.alpha); w.pt(P.prod.beta); w.pt(P.prod.delta); for (int i = 0; i < 5; i++) w.fr(P.prod.z[i]);
    w.pt(P.eq1.alpha); w.fr(P.eq1.z);
    w.sc(P.sc2);
    w.pt(P.comm_vars_at_ry);
    w.dplog(P.polyeval);
    w.pt(P.eq2.alpha); w.fr(P.eq2.z);
}
void read_r1cs(Reader &r, NizkProof &P) {
    P.comm_vars = r.pts((size_t)1 << 20); P.sc1 = r.sc();
    for (int i = 0; i < 4; i++) P.claims_phase2[i] = r.pt();
    P.pok.alpha = r.pt(); P.pok.z1 = r.fr(); P.pok.z2 = r.fr();
    P.prod.alpha = r.pt(); P.prod.beta = r.pt(); P.prod.delta = r.pt(); for (int i = 0; i < 5; i++) P.prod.z[i] = r.fr();
    P.eq1.alpha = r.pt(); P.eq1.z = r.fr();
    P.sc2 = r.sc();
    P.comm_vars_at_ry = r.pt();
    P.polyeval = r.dplog();
    P.eq2.alpha = r.pt(); P.eq2.z = r.fr();
}
}  // namespace

std::vector<uint8_t> CompComm::serialize() const {
    Writer w;
    w.u64(num_cons); w.u64(num_vars); w.u64(num_inputs); w.u64(batch_size); w.u64(num_ops); w.u64(num_mem_cells);
    w.pts(comm_ops); w.pts(comm_mem);
    return std::move(w.b);
}
std::unique_ptr<CompComm> CompComm::parse(const uint8_t *p, size_t n) {
    Reader r{p, n}; auto c = std::make_unique<CompComm>();
    c->num_cons = r.u64(); c->num_vars = r.u64(); c->num_inputs = r.u64(); c->batch_size = r.u64(); c->num_ops = r.u64(); c->num_mem_cells = r.u64();
    c->comm_ops = r.pts((size_t)1 << 24); c->comm_mem = r.pts((size_t)1 << 24);
    if (r.pos != n || c->batch_size != 3) throw Error(OTTI_ERR_MALFORMED_PROOF, "malformed computation commitment");
    return c;
}
std::vector<uint8_t> SnarkProof::serialize() const {
    Writer w;
    write_r1cs(w, r1cs);
    for (int k = 0; k < 3; k++) w.fr(inst_evals[k]);
    const EvalProof &E = eval;
    w.pts(E.comm_derefs);
    w.evals4(E.eval_row); w.evals4(E.eval_col); w.frs(E.dotp_left, 3); w.frs(E.dotp_right, 3);
    w.batch(E.proof_mem); w.batch(E.proof_ops);
    w.frs(E.h_row_addr, 3); w.frs(E.h_row_read_ts, 3); w.fr(E.h_row_audit);
    w.frs(E.h_col_addr, 3); w.frs(E.h_col_read_ts, 3); w.fr(E.h_col_audit);
    w.frs(E.h_val, 3); w.frs(E.h_deref_row, 3); w.frs(E.h_deref_col, 3);
    w.dplog(E.pe_ops); w.dplog(E.pe_mem); w.dplog(E.pe_derefs);
    return std::move(w.b);
}
SnarkProof SnarkProof::parse(const uint8_t *p, size_t n) {
    Reader r{p, n}; SnarkProof S;
    read_r1cs(r, S.r1cs);
    for (int k = 0; k < 3; k++) S.inst_evals[k] = r.fr();
    EvalProof &E = S.eval;
    E.comm_derefs = r.pts((size_t)1 << 24);
    r.evals4(E.eval_row); r.evals4(E.eval_col); r.frs_fixed(E.dotp_left, 3); r.frs_fixed(E.dotp_right, 3);
    E.proof_mem = r.batch(); E.proof_ops = r.batch();
    r.frs_fixed(E.h_row_addr, 3); r.frs_fixed(E.h_row_read_ts, 3); E.h_row_audit = r.fr();
    r.frs_fixed(E.h_col_addr, 3); r.frs_fixed(E.h_col_read_ts, 3); E.h_col_audit = r.fr();
    r.frs_fixed(E.h_val, 3); r.frs_fixed(E.h_deref_row, 3); r.frs_fixed(E.h_deref_col, 3);
    E.pe_ops = r.dplog(); E.pe_mem = r.dplog(); E.pe_derefs = r.dplog();
    if (r.pos != n) throw Error(OTTI_ERR_MALFORMED_PROOF, "trailing bytes after proof");
    return S;
}

// ================================================================================================ verifier
namespace {
void append_u64(Transcript &tr, const char *label, uint64_t x) { uint8_t b[8]; for (int i = 0; i < 8; i++) { b[i] = (uint8_t)x; x >>= 8; } tr.append_message(label, b, 8); }
void append_poly_commitment(Transcript &tr, const char *label, const std::vector<CPoint> &C) {
    tr.append_message(label, "poly_commitment_begin", 21);
    for (auto &c : C) tr.append_point("poly_commitment_share", c.b);
    tr.append_message(label, "poly_commitment_end", 19);
}
void append_unipoly(Transcript &tr, const Fr *c, size_t n) {
    tr.append_message("poly", "UniPoly_begin", 13);
    for (size_t i = 0; i < n; i++) tr.append_scalar("coeff", c[i]);
    tr.append_message("poly", "UniPoly_end", 11);
}
void append_evals4(Transcript &tr, const Evals4 &e, bool col) {
    tr.append_scalar(col ? "claim_col_eval_init" : "claim_row_eval_init", e.init);
    tr.append_scalars(col ? "claim_col_eval_read" : "claim_row_eval_read", e.read, 3);
    tr.append_scalars(col ? "claim_col_eval_write" : "claim_row_eval_write", e.write, 3);
    tr.append_scalar(col ? "claim_col_eval_audit" : "claim_row_eval_audit", e.audit);
}
// the n-to-1 reduction of claimed evaluations: bound_poly_var_bot with the last challenge first
Fr reduce_evals(std::vector<Fr> v, const std::vector<Fr> &ch) {
    for (size_t i = ch.size(); i-- > 0;) { size_t h = v.size() / 2; for (size_t k = 0; k < h; k++) v[k] = fr_add(v[2 * k], fr_mul(ch[i], fr_sub(v[2 * k + 1], v[2 * k]))); v.resize(h); }
    return v[0];
}
Fr hash3(const Fr &addr, const Fr &val, const Fr &ts, const Fr &r, const Fr &r2, const Fr &gamma) { return fr_sub(fr_add(fr_add(fr_mul(ts, r2), fr_mul(val, r)), addr), gamma); }
// SumcheckInstanceProof::verify (degree 3, compressed polynomials)
Fr sc_verify(const std::vector<Fr> &coeffs, Fr e, size_t rounds, Transcript &tr, std::vector<Fr> &r) {
    require(coeffs.size() == 3 * rounds); r.clear();
    for (size_t i = 0; i < rounds; i++) {
        Fr poly[4] = {coeffs[3 * i], fr_zero(), coeffs[3 * i + 1], coeffs[3 * i + 2]};
        poly[1] = fr_sub(fr_sub(fr_sub(fr_sub(e, poly[0]), poly[0]), poly[2]), poly[3]);      // CompressedUniPoly::decompress
        append_unipoly(tr, poly, 4);
        Fr r_i = tr.challenge_scalar("challenge_nextround"); r.push_back(r_i);
        e = unipoly_eval(poly, 4, r_i);
    }
    return e;
}
// ProductCircuitEvalProofBatched::verify
void pcbatch_verify(const ProductCircuitEvalProofBatched &pf, const std::vector<Fr> &claims_prod, const std::vector<Fr> &claims_dotp, size_t len, Transcript &tr,
                    std::vector<Fr> &claims_out, std::vector<Fr> &dotp_out, std::vector<Fr> &rand) {
    const size_t nl = std::max<size_t>(1, ilog2(len)), np = claims_prod.size(), nd = claims_dotp.size();
    require(pf.layers.size() == nl && (nd == 0 || pf.dotp_left.size() == nd));
    std::vector<Fr> claims = claims_prod, rprod; rand.clear(); dotp_out.clear();
    for (size_t i = 0; i < nl; i++) {
        const LayerProofBatched &L = pf.layers[i];
        require(L.left.size() == np && L.right.size() == np);
        const bool last = i == nl - 1;
        if (last) claims.insert(claims.end(), claims_dotp.begin(), claims_dotp.end());
        std::vector<Fr> coeff = tr.challenge_vector("rand_coeffs_next_layer", claims.size());
        Fr claim = fr_zero(); for (size_t k = 0; k < claims.size(); k++) claim = fr_add(claim, fr_mul(claims[k], coeff[k]));
        const Fr claim_last = sc_verify(L.coeffs, claim, i, tr, rprod);
        for (size_t k = 0; k < np; k++) { tr.append_scalar("claim_prod_left", L.left[k]); tr.append_scalar("claim_prod_right", L.right[k]); }
        Fr eq = fr_one(); const Fr one = fr_one();
        for (size_t k = 0; k < rand.size(); k++) eq = fr_mul(eq, fr_add(fr_mul(rand[k], rprod[k]), fr_mul(fr_sub(one, rand[k]), fr_sub(one, rprod[k]))));
        Fr expected = fr_zero();
        for (size_t k = 0; k < np; k++) expected = fr_add(expected, fr_mul(coeff[k], fr_mul(fr_mul(L.left[k], L.right[k]), eq)));
        if (last) for (size_t k = 0; k < nd; k++) {
            tr.append_scalar("claim_dotp_left", pf.dotp_left[k]); tr.append_scalar("claim_dotp_right", pf.dotp_right[k]); tr.append_scalar("claim_dotp_weight", pf.dotp_weight[k]);
            expected = fr_add(expected, fr_mul(coeff[np + k], fr_mul(fr_mul(pf.dotp_left[k], pf.dotp_right[k]), pf.dotp_weight[k])));
        }
        require(fr_eq(expected, claim_last));
        const Fr r_layer = tr.challenge_scalar("challenge_r_layer");
        claims.resize(np);
        for (size_t k = 0; k < np; k++) claims[k] = fr_add(L.left[k], fr_mul(r_layer, fr_sub(L.right[k], L.left[k])));
        if (last) for (size_t k = 0; k < nd / 2; k++)            // the two halves of a dot-product circuit recombine
            for (const std::vector<Fr> *src : {&pf.dotp_left, &pf.dotp_right, &pf.dotp_weight}) dotp_out.push_back(fr_add((*src)[2 * k], fr_mul(r_layer, fr_sub((*src)[2 * k + 1], (*src)[2 * k]))));
        std::vector<Fr> ext = {r_layer}; ext.insert(ext.end(), rprod.begin(), rprod.end()); rand = ext;
    }
    claims_out = claims;
}
// PolyEvalProof::verify_plain: the commitment opens to Zr (blind zero) at r
void polyeval_verify_plain(const DotProductProofLog &pf, const PcSet &s, const Gens &g, const std::vector<Fr> &r, const Fr &Zr, const std::vector<CPoint> &comm, RowSum &rows, Transcript &tr,
                           Deferred &later) {
    require(r.size() == s.num_vars && comm.size() == s.L && rows.n == s.L && rows.C == comm.data() && pf.L_vec.size() == ilog2(s.R));
    const auto t_b = std::chrono::steady_clock::now();
    CPoint C_Zr; { Term t = {s.g1, Zr}; g.commit_terms_c(C_Zr.b, &t, 1); }
    if (getenv("OTTI_TRACE")) fprintf(stderr, "[otti]   polyeval_verify_plain: C_Zr %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_b).count());
    tr.append_protocol_name("polynomial evaluation proof");
    const size_t lv = s.num_vars / 2;
    std::vector<Fr> Lv = eq_evals_host(r.data(), lv), Rv = eq_evals_host(r.data() + lv, s.num_vars - lv);
    static const bool trace = getenv("OTTI_TRACE") != nullptr; const auto t_a = std::chrono::steady_clock::now();
    CPoint C_LZ; pt_encode(C_LZ.b, rows.finish(Lv.data()));
    if (trace) fprintf(stderr, "[otti]   polyeval_verify_plain: C_Zr + eq tables + row sum %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count());
    const PcView pv = {s.h_n, s.g1, s.h1, s.R};
    dotproductlog_verify(pf, s.R, g, pv, tr, Rv.data(), C_LZ, C_Zr, &later);
}
// HashLayerProof::verify_helper
void hash_verify_helper(const std::vector<Fr> &rand_mem, const Evals4 &claims, const Fr ops_val[3], const Fr ops_addr[3], const Fr read_ts[3], const Fr &audit_ts,
                        const std::vector<Fr> &r, const Fr &r_hash, const Fr &gamma) {
    const Fr r2 = fr_mul(r_hash, r_hash), one = fr_one(); const size_t nm = rand_mem.size();
    Fr init_addr = fr_zero(), init_val = fr_one();
    for (size_t i = 0; i < nm; i++) {                                 // IdentityPolynomial / EqPolynomial(r) evaluated at rand_mem
        init_addr = fr_add(init_addr, fr_mul(fr_from_u64((uint64_t)1 << (nm - i - 1)), rand_mem[i]));
        init_val = fr_mul(init_val, fr_add(fr_mul(r[i], rand_mem[i]), fr_mul(fr_sub(one, r[i]), fr_sub(one, rand_mem[i]))));
    }
    require(fr_eq(hash3(init_addr, init_val, fr_zero(), r_hash, r2, gamma), claims.init));
    for (int k = 0; k < 3; k++) {
        require(fr_eq(hash3(ops_addr[k], ops_val[k], read_ts[k], r_hash, r2, gamma), claims.read[k]));
        require(fr_eq(hash3(ops_addr[k], ops_val[k], fr_add(read_ts[k], one), r_hash, r2, gamma), claims.write[k]));
    }
    require(fr_eq(hash3(init_addr, init_val, audit_ts, r_hash, r2, gamma), claims.audit));
}
// R1CSEvalProof::verify
void evalproof_verify(const EvalProof &E, const CompComm &c, const std::vector<Fr> &rx, const std::vector<Fr> &ry, const Fr evals[3], const SnarkGens &g, Transcript &tr) {
    const bool trace = getenv("OTTI_TRACE") != nullptr; auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (!trace) return; const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[otti] evalproof_verify %-32s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_lap).count()); t_lap = t; };
    tr.append_protocol_name("Sparse polynomial evaluation proof");
    // the three polynomial commitments whose rows the closing evaluation proofs sum are known now: with a device they decompress while
    // the host verifies the layered sum-checks
    RowSum rows_derefs(E.comm_derefs.data(), E.comm_derefs.size()), rows_ops(c.comm_ops.data(), c.comm_ops.size()), rows_mem(c.comm_mem.data(), c.comm_mem.size());
    Deferred later;                                                   // the closing group equations of the three evaluation proofs run beside the rest
    const size_t nm = std::max(rx.size(), ry.size()), N = c.num_ops, M = c.num_mem_cells;
    std::vector<Fr> rxe(nm - rx.size(), fr_zero()), rye(nm - ry.size(), fr_zero());     // equalize: zeros in FRONT of the shorter point
    rxe.insert(rxe.end(), rx.begin(), rx.end()); rye.insert(rye.end(), ry.begin(), ry.end());
    require(((size_t)1 << nm) == M && N >= 2 && N == next_pow2(N));
    tr.append_message("derefs_commitment", "begin_derefs_commitment", 23);
    append_poly_commitment(tr, "comm_poly_row_col_ops_val", E.comm_derefs);
    tr.append_message("derefs_commitment", "end_derefs_commitment", 21);
    std::vector<Fr> r_mem_check = tr.challenge_vector("challenge_r_hash", 2);
    tr.append_protocol_name("Sparse polynomial evaluation proof");
    // ---- ProductLayerProof::verify
    tr.append_protocol_name("Sparse polynomial product layer proof");
    for (int side = 0; side < 2; side++) {                            // subset check: init * prod(writes) == prod(reads) * audit
        const Evals4 &e = side ? E.eval_col : E.eval_row;
        Fr ws = fr_one(), rs = fr_one();
        for (int k = 0; k < 3; k++) { ws = fr_mul(ws, e.write[k]); rs = fr_mul(rs, e.read[k]); }
        require(fr_eq(fr_mul(e.init, ws), fr_mul(rs, e.audit)));
        append_evals4(tr, e, side == 1);
    }
    std::vector<Fr> claims_dotp_circuit;
    for (int k = 0; k < 3; k++) {
        require(fr_eq(fr_add(E.dotp_left[k], E.dotp_right[k]), evals[k]));
        tr.append_scalar("claim_eval_dotp_left", E.dotp_left[k]); tr.append_scalar("claim_eval_dotp_right", E.dotp_right[k]);
        claims_dotp_circuit.push_back(E.dotp_left[k]); claims_dotp_circuit.push_back(E.dotp_right[k]);
    }
    std::vector<Fr> claims_prod(12), claims_ops, claims_dotp, rand_ops, claims_mem, none, rand_mem;
    for (int k = 0; k < 3; k++) { claims_prod[k] = E.eval_row.read[k]; claims_prod[3 + k] = E.eval_row.write[k]; claims_prod[6 + k] = E.eval_col.read[k]; claims_prod[9 + k] = E.eval_col.write[k]; }
    lap("commitments appended, jobs begun");
    pcbatch_verify(E.proof_ops, claims_prod, claims_dotp_circuit, N, tr, claims_ops, claims_dotp, rand_ops);
    lap("batched layered sum-check: ops");
    pcbatch_verify(E.proof_mem, {E.eval_row.init, E.eval_row.audit, E.eval_col.init, E.eval_col.audit}, {}, M, tr, claims_mem, none, rand_mem);
    lap("batched layered sum-check: mem");
    require(claims_dotp.size() == 9 && rand_mem.size() == nm);
    // ---- HashLayerProof::verify
    tr.append_protocol_name("Sparse polynomial hash layer proof");
    auto joint = [&](std::vector<Fr> ev, const char *label_evals, const char *label_ch, const char *label_joint, const std::vector<Fr> &rand, std::vector<Fr> &r_joint) {
        tr.append_scalars(label_evals, ev.data(), ev.size());
        std::vector<Fr> ch = tr.challenge_vector(label_ch, ilog2(ev.size()));
        const Fr j = reduce_evals(ev, ch);
        r_joint = ch; r_joint.insert(r_joint.end(), rand.begin(), rand.end());
        tr.append_scalar(label_joint, j);
        return j;
    };
    {   // DerefsEvalProof::verify
        tr.append_protocol_name("Derefs evaluation proof");
        std::vector<Fr> ev(8, fr_zero()), rj;
        for (int k = 0; k < 3; k++) { ev[k] = E.h_deref_row[k]; ev[3 + k] = E.h_deref_col[k]; }
        const Fr j = joint(ev, "evals_ops_val", "challenge_combine_n_to_one", "joint_claim_eval", rand_ops, rj);
        polyeval_verify_plain(E.pe_derefs, g.derefs, *g.eval, rj, j, E.comm_derefs, rows_derefs, tr, later);
        lap("polyeval derefs");
    }
    for (int k = 0; k < 3; k++) require(fr_eq(claims_dotp[3 * k], E.h_deref_row[k]) && fr_eq(claims_dotp[3 * k + 1], E.h_deref_col[k]) && fr_eq(claims_dotp[3 * k + 2], E.h_val[k]));
    {
        std::vector<Fr> ev(16, fr_zero()), rj;
        for (int k = 0; k < 3; k++) { ev[k] = E.h_row_addr[k]; ev[3 + k] = E.h_row_read_ts[k]; ev[6 + k] = E.h_col_addr[k]; ev[9 + k] = E.h_col_read_ts[k]; ev[12 + k] = E.h_val[k]; }
        const Fr j = joint(ev, "claim_evals_ops", "challenge_combine_n_to_one", "joint_claim_eval_ops", rand_ops, rj);
        polyeval_verify_plain(E.pe_ops, g.ops, *g.eval, rj, j, c.comm_ops, rows_ops, tr, later);
        lap("polyeval ops");
    }
    {
        std::vector<Fr> rj;
        const Fr j = joint({E.h_row_audit, E.h_col_audit}, "claim_evals_mem", "challenge_combine_two_to_one", "joint_claim_eval_mem", rand_mem, rj);
        polyeval_verify_plain(E.pe_mem, g.mem, *g.eval, rj, j, c.comm_mem, rows_mem, tr, later);
        lap("polyeval mem");
    }
    Evals4 crow, ccol;                                                // the product layer's claims at (rand_mem, rand_ops)
    crow.init = claims_mem[0]; crow.audit = claims_mem[1]; ccol.init = claims_mem[2]; ccol.audit = claims_mem[3];
    for (int k = 0; k < 3; k++) { crow.read[k] = claims_ops[k]; crow.write[k] = claims_ops[3 + k]; ccol.read[k] = claims_ops[6 + k]; ccol.write[k] = claims_ops[9 + k]; }
    hash_verify_helper(rand_mem, crow, E.h_deref_row, E.h_row_addr, E.h_row_read_ts, E.h_row_audit, rxe, r_mem_check[0], r_mem_check[1]);
    hash_verify_helper(rand_mem, ccol, E.h_deref_col, E.h_col_addr, E.h_col_read_ts, E.h_col_audit, rye, r_mem_check[0], r_mem_check[1]);
    later.finish();
}
}  // namespace

void snark_append_comm(Transcript &tr, const CompComm &c) {          // R1CSCommitment / SparseMatPolyCommitment::append_to_transcript
    append_u64(tr, "num_cons", c.num_cons); append_u64(tr, "num_vars", c.num_vars); append_u64(tr, "num_inputs", c.num_inputs);
    append_u64(tr, "batch_size", c.batch_size); append_u64(tr, "num_ops", c.num_ops); append_u64(tr, "num_mem_cells", c.num_mem_cells);
    append_poly_commitment(tr, "comm_comb_ops", c.comm_ops);
    append_poly_commitment(tr, "comm_comb_mem", c.comm_mem);
}

int snark_verify(const CompComm &comm, const std::vector<Fr> &inputs, const SnarkGens &g, const void *tlabel, size_t tlabel_len, const uint8_t *proof, size_t proof_len) {
    try {
        if (inputs.size() != comm.num_inputs) return OTTI_ERR_INVALID_NUM_INPUTS;
        if (comm.num_cons != g.num_cons || comm.num_vars != g.num_vars) return OTTI_ERR_VERIFY_INTERNAL;
        SnarkProof S = SnarkProof::parse(proof, proof_len);
        Transcript tr(tlabel, tlabel_len);
        tr.append_protocol_name("Spartan SNARK proof");
        snark_append_comm(tr, comm);
        std::vector<Fr> rx, ry;
        if (int rc = r1cs_verify_host(S.r1cs, comm.num_cons, comm.num_vars, inputs, S.inst_evals, *g.sat, tr, rx, ry)) return rc;
        tr.append_scalar("Ar_claim", S.inst_evals[0]); tr.append_scalar("Br_claim", S.inst_evals[1]); tr.append_scalar("Cr_claim", S.inst_evals[2]);
        evalproof_verify(S.eval, comm, rx, ry, S.inst_evals, g, tr);
        return OTTI_OK;
    } catch (const VerifyFail &f) { return f.code; }
    catch (const Error &e) { return e.code; }
}

}  // namespace otti
