/*
 * ORACLE — test infrastructure only (see fr.h).  CPU restatement of libspartan's SNARK mode: the computation commitment
 * (SNARK::encode), R1CSEvalProof (sparse-polynomial evaluation proof: memory-checking product circuits, the batched non-ZK cubic
 * sum-check, hash layer) and SNARK::{prove, verify} — what `spzk verify` without `--nizk` reaches
 * [REF /root/reference/run.py:58,100 invoke the binary with --nizk only; BASELINE.json's metric names the SNARK].
 *
 * PARITY UNPINNED, as for spartan.c: restated from upstream microsoft/Spartan [RECALL: src/sparse_mlpoly.rs, src/product_tree.rs,
 * src/sumcheck.rs (SumcheckInstanceProof::prove_cubic_batched), src/unipoly.rs (CompressedUniPoly), src/r1csinstance.rs
 * (R1CSCommitment, R1CSEvalProof), src/lib.rs (SNARKGens, SNARK)]; the reference's Spartan/ submodule is empty
 * (/root/reference/.gitmodules:4-6).  Every function names the upstream item it follows.
 */
#include "internal.h"
#include <stdlib.h>
#include <string.h>
#include <assert.h>

/* ------------------------------------------------------------------ dense_mlpoly.rs PolyCommitmentGens */
typedef struct { orc_mcgens gens_n, gens_1; } pcgens_t;              /* DotProductProofGens {n, gens_n, gens_1} */
struct orc_snark_gens {
    orc_gens *sat;                                                    /* gens_r1cs_sat */
    pcgens_t ops, mem, derefs;                                        /* gens_r1cs_eval: SparseMatPolyCommitmentGens */
    size_t vars_ops, vars_mem, vars_derefs;
};
static void pcgens_new(pcgens_t *g, const ge_t *stream, size_t num_vars) {
    size_t R = (size_t)1 << (num_vars - num_vars / 2);               /* compute_factored_lens: right = ell - ell/2 */
    mcgens_from(&g->gens_n, stream, R, &stream[R + 1]);
    mcgens_from(&g->gens_1, &stream[R], 1, &stream[R + 1]);
}
static size_t log2_pow2(size_t n) { return ilog2(next_pow2(n)); }

/* lib.rs SNARKGens::new -> R1CSCommitmentGens::new -> SparseMatPolyCommitmentGens::new(label, x, y, nz, batch_size = 3) */
orc_snark_gens *orc_snark_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs, size_t num_nz_entries) {
    orc_snark_gens *g = (orc_snark_gens *)calloc(1, sizeof *g);
    size_t nvp = num_vars > num_inputs + 1 ? num_vars : num_inputs + 1; nvp = next_pow2(nvp);
    size_t ncp = num_cons < 2 ? 2 : next_pow2(num_cons);
    g->sat = orc_gens_new(num_cons, num_vars, num_inputs);
    size_t vx = ilog2(ncp), vy = ilog2(2 * nvp), lgnz = log2_pow2(num_nz_entries > 2 ? num_nz_entries : 2);   /* at least two operations per matrix */
    g->vars_ops = lgnz + log2_pow2(3 * 5);
    g->vars_mem = (vx > vy ? vx : vy) + 1;
    g->vars_derefs = lgnz + log2_pow2(3 * 2);
    size_t vmax = g->vars_ops > g->vars_mem ? g->vars_ops : g->vars_mem; if (g->vars_derefs > vmax) vmax = g->vars_derefs;
    size_t count = ((size_t)1 << (vmax - vmax / 2)) + 2;
    ge_t *P = (ge_t *)malloc(count * sizeof(ge_t));
    gens_stream(P, count, "gens_r1cs_eval");                           /* the three generator sets are prefixes of one stream */
    pcgens_new(&g->ops, P, g->vars_ops); pcgens_new(&g->mem, P, g->vars_mem); pcgens_new(&g->derefs, P, g->vars_derefs);
    free(P);
    return g;
}
static void pcgens_free(pcgens_t *g) { mcgens_free(&g->gens_n); mcgens_free(&g->gens_1); }
void orc_snark_gens_free(orc_snark_gens *g) { if (!g) return; orc_gens_free(g->sat); pcgens_free(&g->ops); pcgens_free(&g->mem); pcgens_free(&g->derefs); free(g); }
const orc_gens *orc_snark_gens_sat(const orc_snark_gens *g) { return g->sat; }

/* ------------------------------------------------------------------ dense_mlpoly.rs helpers */
/* DensePolynomial::commit(gens, None): rows of the L x R matrix, zero blinds */
static uint8_t *poly_commit(const fr_t *Z, size_t num_vars, const pcgens_t *g, size_t *rows_out) {
    size_t L = (size_t)1 << (num_vars / 2), R = (size_t)1 << (num_vars - num_vars / 2);
    fr_t *blinds = (fr_t *)calloc(L, sizeof(fr_t));
    uint8_t *C = (uint8_t *)malloc(32 * L);
    orc_commit_rows(Z, L, R, blinds, &g->gens_n, C);
    free(blinds); *rows_out = L; return C;
}
/* PolyCommitment::append_to_transcript */
static void append_poly_commitment(transcript_t *tr, const char *label, const uint8_t *C, size_t rows) {
    tr_append(tr, label, (const uint8_t *)"poly_commitment_begin", 21);
    for (size_t i = 0; i < rows; i++) tr_append_point(tr, "poly_commitment_share", C + 32 * i);
    tr_append(tr, label, (const uint8_t *)"poly_commitment_end", 19);
}
static void append_u64(transcript_t *tr, const char *label, uint64_t x) { uint8_t b[8]; for (int i = 0; i < 8; i++) { b[i] = (uint8_t)x; x >>= 8; } tr_append(tr, label, b, 8); }
/* DensePolynomial::evaluate */
static void poly_evaluate(fr_t *o, const fr_t *Z, size_t num_vars, const fr_t *r) {
    size_t n = (size_t)1 << num_vars;
    fr_t *chis = (fr_t *)malloc(n * sizeof(fr_t));
    orc_eq_evals(r, num_vars, chis); dot(o, Z, chis, n); free(chis);
}
/* the n-to-1 reduction used by the hash layer: fold `evals` (a power of two of them) with bound_poly_var_bot, last challenge first */
static void reduce_evals(fr_t *out, const fr_t *evals, size_t n, const fr_t *challenges) {
    fr_t *v = (fr_t *)malloc(n * sizeof(fr_t)); memcpy(v, evals, n * sizeof(fr_t));
    size_t len = n, k = ilog2(n);
    for (size_t i = k; i-- > 0;) { orc_fold_bot(v, len, &challenges[i]); len /= 2; }
    *out = v[0]; free(v);
}
/* PolyEvalProof::prove(poly, None, r, Zr, None, gens, ..) */
static void polyeval_prove(dplog_t *pf, const fr_t *Z, size_t num_vars, const fr_t *r, const fr_t *Zr, const pcgens_t *g, transcript_t *tr, transcript_t *tape) {
    tr_protocol_name(tr, "polynomial evaluation proof");
    size_t lv = num_vars / 2, L = (size_t)1 << lv, R = (size_t)1 << (num_vars - lv);
    fr_t *Lv = (fr_t *)malloc(L * sizeof(fr_t)), *Rv = (fr_t *)malloc(R * sizeof(fr_t)), *LZ = (fr_t *)malloc(R * sizeof(fr_t));
    orc_eq_evals(r, lv, Lv); orc_eq_evals(r + lv, num_vars - lv, Rv);
    orc_poly_bound(Z, L, R, Lv, LZ);
    uint8_t Cy[32];
    dplog_prove(pf, Cy, &g->gens_n, &g->gens_1, tr, tape, LZ, &FR_ZERO, Rv, R, Zr, &FR_ZERO);
    free(Lv); free(Rv); free(LZ);
}
/* PolyEvalProof::verify_plain(gens, transcript, r, Zr, comm) */
static int polyeval_verify_plain(const dplog_t *pf, size_t num_vars, const fr_t *r, const fr_t *Zr, const uint8_t *comm, size_t rows, const pcgens_t *g, transcript_t *tr) {
    size_t lv = num_vars / 2, L = (size_t)1 << lv, R = (size_t)1 << (num_vars - lv);
    if (rows != L || pf->n != ilog2(R)) return ORC_ERR_VERIFY_INTERNAL;
    uint8_t C_Zr[32]; commit_scalar_c(C_Zr, Zr, &FR_ZERO, &g->gens_1);
    tr_protocol_name(tr, "polynomial evaluation proof");
    fr_t *Lv = (fr_t *)malloc(L * sizeof(fr_t)), *Rv = (fr_t *)malloc(R * sizeof(fr_t));
    orc_eq_evals(r, lv, Lv); orc_eq_evals(r + lv, num_vars - lv, Rv);
    ge_t *Cs = (ge_t *)malloc(L * sizeof(ge_t)); int rc = ORC_OK;
    for (size_t i = 0; i < L && !rc; i++) if (!ge_decode(&Cs[i], comm + 32 * i)) rc = ORC_ERR_VERIFY_DECOMPRESS;
    if (!rc) { ge_t CLZ; uint8_t C_LZ[32]; ge_msm(&CLZ, Lv, Cs, L); ge_encode(C_LZ, &CLZ); rc = dplog_verify(pf, R, &g->gens_n, &g->gens_1, tr, Rv, C_LZ, C_Zr); }
    free(Lv); free(Rv); free(Cs);
    return rc;
}

/* ------------------------------------------------------------------ sparse_mlpoly.rs: dense representation + commitment */
typedef struct {
    size_t N, M;                                                      /* num_ops (per matrix, padded), num_mem_cells */
    size_t *row_addr[3], *col_addr[3];
    fr_t *val[3], *row_addr_f[3], *row_read_ts[3], *col_addr_f[3], *col_read_ts[3];
    fr_t *row_audit_ts, *col_audit_ts;
    fr_t *comb_ops, *comb_mem; size_t vars_comb_ops, vars_comb_mem;
} dense_rep_t;
struct orc_snark_comm {
    size_t num_cons, num_vars, num_inputs;                            /* R1CSCommitment */
    size_t batch_size, num_ops, num_mem_cells;                        /* SparseMatPolyCommitment */
    uint8_t *comm_ops, *comm_mem; size_t rows_ops, rows_mem;
    dense_rep_t d;                                                    /* R1CSDecommitment (prover side only) */
};
/* AddrTimestamps::new: read_ts[i] = audit_ts[addr] before the op, audit_ts[addr] += 1; audit_ts is shared by the three matrices */
static void addr_timestamps(size_t M, size_t N, size_t *const addr[3], fr_t *addr_f[3], fr_t *read_ts[3], fr_t **audit_out) {
    size_t *audit = (size_t *)calloc(M, sizeof(size_t));
    for (int k = 0; k < 3; k++) {
        addr_f[k] = (fr_t *)malloc(N * sizeof(fr_t)); read_ts[k] = (fr_t *)malloc(N * sizeof(fr_t));
        for (size_t i = 0; i < N; i++) {
            size_t a = addr[k][i]; assert(a < M);
            fr_from_u64(&addr_f[k][i], a); fr_from_u64(&read_ts[k][i], audit[a]); audit[a]++;
        }
    }
    fr_t *af = (fr_t *)malloc(M * sizeof(fr_t));
    for (size_t i = 0; i < M; i++) fr_from_u64(&af[i], audit[i]);
    free(audit); *audit_out = af;
}
/* SparseMatPolynomial::multi_sparse_to_dense_rep + DensePolynomial::merge */
static void dense_rep_build(dense_rep_t *d, const orc_instance *I) {
    const orc_sparse *m[3] = {&I->A, &I->B, &I->C};
    size_t nz = 0; for (int k = 0; k < 3; k++) if (m[k]->n > nz) nz = m[k]->n;
    size_t N = next_pow2(nz > 2 ? nz : 2), vx = ilog2(I->num_cons), vy = ilog2(2 * I->num_vars);
    size_t M = (size_t)1 << (vx > vy ? vx : vy);
    memset(d, 0, sizeof *d); d->N = N; d->M = M;
    for (int k = 0; k < 3; k++) {                                     /* sparse_to_dense_vecs: padded with (row 0, col 0, value 0) */
        d->row_addr[k] = (size_t *)calloc(N, sizeof(size_t)); d->col_addr[k] = (size_t *)calloc(N, sizeof(size_t)); d->val[k] = (fr_t *)calloc(N, sizeof(fr_t));
        for (size_t i = 0; i < m[k]->n; i++) { d->row_addr[k][i] = m[k]->M[i].row; d->col_addr[k][i] = m[k]->M[i].col; d->val[k][i] = m[k]->M[i].val; }
    }
    addr_timestamps(M, N, d->row_addr, d->row_addr_f, d->row_read_ts, &d->row_audit_ts);
    addr_timestamps(M, N, d->col_addr, d->col_addr_f, d->col_read_ts, &d->col_audit_ts);
    /* comb_ops = merge(row.ops_addr, row.read_ts, col.ops_addr, col.read_ts, val), zero-padded to a power of two; comb_mem = row.audit_ts || col.audit_ts */
    size_t total = next_pow2(15 * N);
    d->comb_ops = (fr_t *)calloc(total, sizeof(fr_t)); d->vars_comb_ops = ilog2(total);
    fr_t **parts[5] = {d->row_addr_f, d->row_read_ts, d->col_addr_f, d->col_read_ts, d->val};
    for (int p = 0; p < 5; p++) for (int k = 0; k < 3; k++) memcpy(d->comb_ops + (size_t)(3 * p + k) * N, parts[p][k], N * sizeof(fr_t));
    d->comb_mem = (fr_t *)malloc(2 * M * sizeof(fr_t)); d->vars_comb_mem = ilog2(2 * M);
    memcpy(d->comb_mem, d->row_audit_ts, M * sizeof(fr_t)); memcpy(d->comb_mem + M, d->col_audit_ts, M * sizeof(fr_t));
}
static void dense_rep_free(dense_rep_t *d) {
    for (int k = 0; k < 3; k++) { free(d->row_addr[k]); free(d->col_addr[k]); free(d->val[k]); free(d->row_addr_f[k]); free(d->row_read_ts[k]); free(d->col_addr_f[k]); free(d->col_read_ts[k]); }
    free(d->row_audit_ts); free(d->col_audit_ts); free(d->comb_ops); free(d->comb_mem);
}
/* lib.rs SNARK::encode -> R1CSInstance::commit -> SparseMatPolynomial::multi_commit */
orc_snark_comm *orc_snark_encode(const orc_instance *I, const orc_snark_gens *g) {
    orc_snark_comm *c = (orc_snark_comm *)calloc(1, sizeof *c);
    c->num_cons = I->num_cons; c->num_vars = I->num_vars; c->num_inputs = I->num_inputs;
    dense_rep_build(&c->d, I);
    c->batch_size = 3; c->num_ops = c->d.N; c->num_mem_cells = c->d.M;
    if (c->d.vars_comb_ops != g->vars_ops || c->d.vars_comb_mem != g->vars_mem) { orc_snark_comm_free(c); return NULL; }   /* generators made for another size */
    c->comm_ops = poly_commit(c->d.comb_ops, c->d.vars_comb_ops, &g->ops, &c->rows_ops);
    c->comm_mem = poly_commit(c->d.comb_mem, c->d.vars_comb_mem, &g->mem, &c->rows_mem);
    return c;
}
void orc_snark_comm_free(orc_snark_comm *c) { if (!c) return; dense_rep_free(&c->d); free(c->comm_ops); free(c->comm_mem); free(c); }
/* bincode of ComputationCommitment { comm: R1CSCommitment { num_cons, num_vars, num_inputs, comm: SparseMatPolyCommitment {..} } } */
void orc_snark_comm_bytes(const orc_snark_comm *c, uint8_t **out, size_t *len) {
    wbuf_t w = {0, 0, 0};
    wb_u64(&w, c->num_cons); wb_u64(&w, c->num_vars); wb_u64(&w, c->num_inputs);
    wb_u64(&w, c->batch_size); wb_u64(&w, c->num_ops); wb_u64(&w, c->num_mem_cells);
    wb_u64(&w, c->rows_ops); wb(&w, c->comm_ops, 32 * c->rows_ops);
    wb_u64(&w, c->rows_mem); wb(&w, c->comm_mem, 32 * c->rows_mem);
    *out = w.p; *len = w.len;
}
/* the verifier's view: the commitment alone, parsed back from those bytes */
orc_snark_comm *orc_snark_comm_parse(const uint8_t *buf, size_t len) {
    rbuf_t r = {buf, len, 0, 0};
    orc_snark_comm *c = (orc_snark_comm *)calloc(1, sizeof *c);
    c->num_cons = rb_u64(&r); c->num_vars = rb_u64(&r); c->num_inputs = rb_u64(&r);
    c->batch_size = rb_u64(&r); c->num_ops = rb_u64(&r); c->num_mem_cells = rb_u64(&r);
    c->comm_ops = rb_vec32(&r, &c->rows_ops, (size_t)1 << 24); c->comm_mem = rb_vec32(&r, &c->rows_mem, (size_t)1 << 24);
    if (r.bad || r.pos != r.len || c->batch_size != 3) { orc_snark_comm_free(c); return NULL; }
    return c;
}
/* R1CSCommitment::append_to_transcript + SparseMatPolyCommitment::append_to_transcript */
static void append_comm(transcript_t *tr, const orc_snark_comm *c) {
    append_u64(tr, "num_cons", c->num_cons); append_u64(tr, "num_vars", c->num_vars); append_u64(tr, "num_inputs", c->num_inputs);
    append_u64(tr, "batch_size", c->batch_size); append_u64(tr, "num_ops", c->num_ops); append_u64(tr, "num_mem_cells", c->num_mem_cells);
    append_poly_commitment(tr, "comm_comb_ops", c->comm_ops, c->rows_ops);
    append_poly_commitment(tr, "comm_comb_mem", c->comm_mem, c->rows_mem);
}

/* ------------------------------------------------------------------ product_tree.rs */
typedef struct { size_t nl; fr_t **left, **right; } pcirc_t;          /* layer k has n / 2^(k+1) elements on each side */
/* ProductCircuit::new */
static void pcirc_new(pcirc_t *c, const fr_t *poly, size_t n) {
    size_t nl = ilog2(n); if (nl == 0) nl = 1;
    c->nl = nl; c->left = (fr_t **)calloc(nl, sizeof(fr_t *)); c->right = (fr_t **)calloc(nl, sizeof(fr_t *));
    size_t h = n / 2;
    c->left[0] = (fr_t *)malloc(h * sizeof(fr_t)); c->right[0] = (fr_t *)malloc(h * sizeof(fr_t));
    memcpy(c->left[0], poly, h * sizeof(fr_t)); memcpy(c->right[0], poly + h, h * sizeof(fr_t));
    for (size_t k = 1; k < nl; k++) {                                 /* compute_layer */
        size_t q = h / 2;
        c->left[k] = (fr_t *)malloc((q ? q : 1) * sizeof(fr_t)); c->right[k] = (fr_t *)malloc((q ? q : 1) * sizeof(fr_t));
#pragma omp parallel for num_threads(g_threads) schedule(static) if (q >= 4096)
        for (size_t i = 0; i < q; i++) { fr_mul(&c->left[k][i], &c->left[k - 1][i], &c->right[k - 1][i]); fr_mul(&c->right[k][i], &c->left[k - 1][q + i], &c->right[k - 1][q + i]); }
        h = q;
    }
}
static void pcirc_eval(fr_t *o, const pcirc_t *c) { fr_mul(o, &c->left[c->nl - 1][0], &c->right[c->nl - 1][0]); }
static void pcirc_free(pcirc_t *c) { for (size_t k = 0; k < c->nl; k++) { free(c->left[k]); free(c->right[k]); } free(c->left); free(c->right); }

/* sumcheck.rs SumcheckInstanceProof (non-ZK): one CompressedUniPoly (coefficients without the linear term: c0, c2, c3) per round */
typedef struct { size_t rounds; fr_t *coeffs; } scproof_t;
typedef struct { scproof_t sc; size_t np; fr_t *left, *right; } layerproof_t;     /* LayerProofBatched */
typedef struct { size_t nlayers; layerproof_t *layers; size_t nd; fr_t *dl, *dr, *dw; } pcbatch_t;   /* ProductCircuitEvalProofBatched */
static void pcbatch_free(pcbatch_t *p) { for (size_t i = 0; i < p->nlayers; i++) { free(p->layers[i].sc.coeffs); free(p->layers[i].left); free(p->layers[i].right); } free(p->layers); free(p->dl); free(p->dr); free(p->dw); }

/* UniPoly::append_to_transcript */
static void append_unipoly(transcript_t *tr, const fr_t *c, size_t n) {
    tr_append(tr, "poly", (const uint8_t *)"UniPoly_begin", 13);
    for (size_t i = 0; i < n; i++) tr_append_scalar(tr, "coeff", &c[i]);
    tr_append(tr, "poly", (const uint8_t *)"UniPoly_end", 11);
}
/* the inner loop of prove_cubic_batched for one (A, B, C) triple: evaluations of sum A*B*C at 0, 2, 3 */
static void cubic_abc_evals(const fr_t *A, const fr_t *B, const fr_t *C, size_t len, fr_t e[3]) {
    fr_t e0 = FR_ZERO, e2 = FR_ZERO, e3 = FR_ZERO;
#pragma omp parallel num_threads(g_threads) if (len >= 2048)
    {
        fr_t l0 = FR_ZERO, l2 = FR_ZERO, l3 = FR_ZERO;
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < len; i++) {
            fr_t t, a, b, c, da, db, dc;
            fr_mul(&t, &A[i], &B[i]); fr_mul(&t, &t, &C[i]); fr_add(&l0, &l0, &t);
            fr_sub(&da, &A[len + i], &A[i]); fr_sub(&db, &B[len + i], &B[i]); fr_sub(&dc, &C[len + i], &C[i]);
            fr_add(&a, &A[len + i], &da); fr_add(&b, &B[len + i], &db); fr_add(&c, &C[len + i], &dc);
            fr_mul(&t, &a, &b); fr_mul(&t, &t, &c); fr_add(&l2, &l2, &t);
            fr_add(&a, &a, &da); fr_add(&b, &b, &db); fr_add(&c, &c, &dc);
            fr_mul(&t, &a, &b); fr_mul(&t, &t, &c); fr_add(&l3, &l3, &t);
        }
#pragma omp critical
        { fr_add(&e0, &e0, &l0); fr_add(&e2, &e2, &l2); fr_add(&e3, &e3, &l3); }
    }
    e[0] = e0; e[1] = e2; e[2] = e3;
}
/* SumcheckInstanceProof::prove_cubic_batched: np (A, B) pairs sharing C_par, then nd (A, B, C) triples; tables are folded in place.
   Returns the round challenges in r; final table values stay in element 0 of every table. */
static void prove_cubic_batched(scproof_t *pf, const fr_t *claim, size_t rounds, fr_t **Ap, fr_t **Bp, size_t np, fr_t *Cpar, fr_t **As, fr_t **Bs, fr_t **Cs, size_t nd,
                                size_t len /* current table length */, const fr_t *coeffs, transcript_t *tr, fr_t *r) {
    pf->rounds = rounds; pf->coeffs = (fr_t *)malloc((3 * rounds + 1) * sizeof(fr_t));
    fr_t e = *claim;
    for (size_t j = 0; j < rounds; j++) {
        size_t h = len / 2;
        fr_t c0 = FR_ZERO, c2 = FR_ZERO, c3 = FR_ZERO, ev[3], t;
        for (size_t k = 0; k < np + nd; k++) {
            if (k < np) cubic_abc_evals(Ap[k], Bp[k], Cpar, h, ev); else cubic_abc_evals(As[k - np], Bs[k - np], Cs[k - np], h, ev);
            fr_mul(&t, &ev[0], &coeffs[k]); fr_add(&c0, &c0, &t); fr_mul(&t, &ev[1], &coeffs[k]); fr_add(&c2, &c2, &t); fr_mul(&t, &ev[2], &coeffs[k]); fr_add(&c3, &c3, &t);
        }
        fr_t evals[4], poly[4], r_j;
        evals[0] = c0; fr_sub(&evals[1], &e, &c0); evals[2] = c2; evals[3] = c3;
        unipoly_from_evals(poly, evals, 4);
        append_unipoly(tr, poly, 4);
        tr_challenge_scalar(tr, "challenge_nextround", &r_j); r[j] = r_j;
        for (size_t k = 0; k < np; k++) { orc_fold_top(Ap[k], len, &r_j); orc_fold_top(Bp[k], len, &r_j); }
        orc_fold_top(Cpar, len, &r_j);
        for (size_t k = 0; k < nd; k++) { orc_fold_top(As[k], len, &r_j); orc_fold_top(Bs[k], len, &r_j); orc_fold_top(Cs[k], len, &r_j); }
        unipoly_eval(&e, poly, 4, &r_j);
        pf->coeffs[3 * j] = poly[0]; pf->coeffs[3 * j + 1] = poly[2]; pf->coeffs[3 * j + 2] = poly[3];      /* UniPoly::compress */
        len = h;
    }
}
/* SumcheckInstanceProof::verify (degree bound 3) */
static int sc_verify(const scproof_t *pf, const fr_t *claim, size_t rounds, transcript_t *tr, fr_t *e_out, fr_t *r) {
    if (pf->rounds != rounds) return ORC_ERR_VERIFY_INTERNAL;
    fr_t e = *claim;
    for (size_t i = 0; i < rounds; i++) {
        fr_t poly[4], lin;                                           /* CompressedUniPoly::decompress(hint = e) */
        poly[0] = pf->coeffs[3 * i]; poly[2] = pf->coeffs[3 * i + 1]; poly[3] = pf->coeffs[3 * i + 2];
        fr_sub(&lin, &e, &poly[0]); fr_sub(&lin, &lin, &poly[0]); fr_sub(&lin, &lin, &poly[2]); fr_sub(&lin, &lin, &poly[3]); poly[1] = lin;
        /* eval_at_zero + eval_at_one == e holds by construction of the linear term */
        append_unipoly(tr, poly, 4);
        tr_challenge_scalar(tr, "challenge_nextround", &r[i]);
        unipoly_eval(&e, poly, 4, &r[i]);
    }
    *e_out = e; return ORC_OK;
}

/* ProductCircuitEvalProofBatched::prove.  circs: np product circuits of equal size; dotp: nd (left, right, weight) triples of the
   length of layer 0 (consumed in place).  rand_out: the final evaluation point (log2 of the circuit size entries). */
static void pcbatch_prove(pcbatch_t *pf, pcirc_t *circs, size_t np, fr_t **dleft, fr_t **dright, fr_t **dweight, size_t nd, size_t dlen, transcript_t *tr, fr_t *rand_out, size_t *nrand_out) {
    size_t nl = circs[0].nl;
    pf->nlayers = nl; pf->layers = (layerproof_t *)calloc(nl, sizeof(layerproof_t)); pf->nd = 0; pf->dl = pf->dr = pf->dw = NULL;
    fr_t *claims = (fr_t *)malloc((np + nd) * sizeof(fr_t)), *coeffs = (fr_t *)malloc((np + nd) * sizeof(fr_t));
    for (size_t i = 0; i < np; i++) pcirc_eval(&claims[i], &circs[i]);
    size_t nclaims = np, nrand = 0;
    fr_t *rand = (fr_t *)malloc((nl + 2) * sizeof(fr_t)), *rprod = (fr_t *)malloc((nl + 2) * sizeof(fr_t));
    fr_t **Ap = (fr_t **)malloc(np * sizeof(fr_t *)), **Bp = (fr_t **)malloc(np * sizeof(fr_t *));
    for (size_t li = 0; li < nl; li++) {
        size_t layer_id = nl - 1 - li, h = (size_t)1 << nrand;       /* elements per side in this layer = 2^(#rand) */
        fr_t *Cpar = (fr_t *)malloc(h * sizeof(fr_t));
        orc_eq_evals(rand, nrand, Cpar);
        for (size_t i = 0; i < np; i++) { Ap[i] = circs[i].left[layer_id]; Bp[i] = circs[i].right[layer_id]; }
        size_t use_d = 0;
        if (layer_id == 0 && nd) {                                    /* the dot-product circuits join at the input layer */
            assert(dlen == h);
            for (size_t i = 0; i < nd; i++) { fr_t acc = FR_ZERO, t; for (size_t x = 0; x < dlen; x++) { fr_mul(&t, &dleft[i][x], &dright[i][x]); fr_mul(&t, &t, &dweight[i][x]); fr_add(&acc, &acc, &t); } claims[nclaims++] = acc; }
            use_d = nd;
        }
        tr_challenge_vector(tr, "rand_coeffs_next_layer", coeffs, nclaims);
        fr_t claim = FR_ZERO, t;
        for (size_t i = 0; i < nclaims; i++) { fr_mul(&t, &claims[i], &coeffs[i]); fr_add(&claim, &claim, &t); }
        layerproof_t *L = &pf->layers[li];
        prove_cubic_batched(&L->sc, &claim, nrand, Ap, Bp, np, Cpar, dleft, dright, dweight, use_d, h, coeffs, tr, rprod);
        L->np = np; L->left = (fr_t *)malloc(np * sizeof(fr_t)); L->right = (fr_t *)malloc(np * sizeof(fr_t));
        for (size_t i = 0; i < np; i++) { L->left[i] = Ap[i][0]; L->right[i] = Bp[i][0]; tr_append_scalar(tr, "claim_prod_left", &L->left[i]); tr_append_scalar(tr, "claim_prod_right", &L->right[i]); }
        if (use_d) {
            pf->nd = nd; pf->dl = (fr_t *)malloc(nd * sizeof(fr_t)); pf->dr = (fr_t *)malloc(nd * sizeof(fr_t)); pf->dw = (fr_t *)malloc(nd * sizeof(fr_t));
            for (size_t i = 0; i < nd; i++) {
                pf->dl[i] = dleft[i][0]; pf->dr[i] = dright[i][0]; pf->dw[i] = dweight[i][0];
                tr_append_scalar(tr, "claim_dotp_left", &pf->dl[i]); tr_append_scalar(tr, "claim_dotp_right", &pf->dr[i]); tr_append_scalar(tr, "claim_dotp_weight", &pf->dw[i]);
            }
        }
        fr_t r_layer; tr_challenge_scalar(tr, "challenge_r_layer", &r_layer);
        for (size_t i = 0; i < np; i++) { fr_sub(&t, &L->right[i], &L->left[i]); fr_mul(&t, &r_layer, &t); fr_add(&claims[i], &L->left[i], &t); }
        nclaims = np;
        rand[0] = r_layer; memcpy(rand + 1, rprod, nrand * sizeof(fr_t)); nrand++;
        free(Cpar);
    }
    memcpy(rand_out, rand, nrand * sizeof(fr_t)); *nrand_out = nrand;
    free(claims); free(coeffs); free(rand); free(rprod); free(Ap); free(Bp);
}
/* ProductCircuitEvalProofBatched::verify: returns the per-circuit claims at the final point, the dot-product claims (left, right, weight
   per ORIGINAL dot-product circuit, i.e. nd / 2 triples) and the point */
static int pcbatch_verify(const pcbatch_t *pf, const fr_t *claims_prod, size_t np, const fr_t *claims_dotp, size_t nd, size_t len, transcript_t *tr,
                          fr_t *claims_out, fr_t *dotp_out, fr_t *rand_out, size_t *nrand_out) {
    size_t nl = ilog2(len); if (nl == 0) nl = 1;
    if (pf->nlayers != nl) return ORC_ERR_VERIFY_INTERNAL;
    if (nd && pf->nd != nd) return ORC_ERR_VERIFY_INTERNAL;
    fr_t *claims = (fr_t *)malloc((np + nd) * sizeof(fr_t)), *coeffs = (fr_t *)malloc((np + nd) * sizeof(fr_t));
    fr_t *rand = (fr_t *)malloc((nl + 2) * sizeof(fr_t)), *rprod = (fr_t *)malloc((nl + 2) * sizeof(fr_t));
    memcpy(claims, claims_prod, np * sizeof(fr_t));
    size_t nclaims = np, nrand = 0; int rc = ORC_OK;
    for (size_t i = 0; i < nl && !rc; i++) {
        const layerproof_t *L = &pf->layers[i];
        if (L->np != np) { rc = ORC_ERR_VERIFY_INTERNAL; break; }
        int last = i == nl - 1;
        if (last) { memcpy(claims + nclaims, claims_dotp, nd * sizeof(fr_t)); nclaims += nd; }
        tr_challenge_vector(tr, "rand_coeffs_next_layer", coeffs, nclaims);
        fr_t claim = FR_ZERO, t, u, claim_last;
        for (size_t k = 0; k < nclaims; k++) { fr_mul(&t, &claims[k], &coeffs[k]); fr_add(&claim, &claim, &t); }
        if ((rc = sc_verify(&L->sc, &claim, i, tr, &claim_last, rprod))) break;
        for (size_t k = 0; k < np; k++) { tr_append_scalar(tr, "claim_prod_left", &L->left[k]); tr_append_scalar(tr, "claim_prod_right", &L->right[k]); }
        fr_t eq = FR_ONE;
        for (size_t k = 0; k < nrand; k++) {
            fr_t om1, om2; fr_mul(&t, &rand[k], &rprod[k]); fr_sub(&om1, &FR_ONE, &rand[k]); fr_sub(&om2, &FR_ONE, &rprod[k]); fr_mul(&u, &om1, &om2); fr_add(&t, &t, &u); fr_mul(&eq, &eq, &t);
        }
        fr_t expected = FR_ZERO;
        for (size_t k = 0; k < np; k++) { fr_mul(&t, &L->left[k], &L->right[k]); fr_mul(&t, &t, &eq); fr_mul(&t, &coeffs[k], &t); fr_add(&expected, &expected, &t); }
        if (last) for (size_t k = 0; k < nd; k++) {
            tr_append_scalar(tr, "claim_dotp_left", &pf->dl[k]); tr_append_scalar(tr, "claim_dotp_right", &pf->dr[k]); tr_append_scalar(tr, "claim_dotp_weight", &pf->dw[k]);
            fr_mul(&t, &pf->dl[k], &pf->dr[k]); fr_mul(&t, &t, &pf->dw[k]); fr_mul(&t, &coeffs[np + k], &t); fr_add(&expected, &expected, &t);
        }
        if (!fr_eq(&expected, &claim_last)) { rc = ORC_ERR_VERIFY_INTERNAL; break; }
        fr_t r_layer; tr_challenge_scalar(tr, "challenge_r_layer", &r_layer);
        for (size_t k = 0; k < np; k++) { fr_sub(&t, &L->right[k], &L->left[k]); fr_mul(&t, &r_layer, &t); fr_add(&claims[k], &L->left[k], &t); }
        nclaims = np;
        if (last) for (size_t k = 0; k < nd / 2; k++) {             /* the two halves of a dot-product circuit recombine */
            const fr_t *src[3] = {pf->dl, pf->dr, pf->dw};
            for (int q = 0; q < 3; q++) { fr_sub(&t, &src[q][2 * k + 1], &src[q][2 * k]); fr_mul(&t, &r_layer, &t); fr_add(&dotp_out[3 * k + q], &src[q][2 * k], &t); }
        }
        rand[0] = r_layer; memcpy(rand + 1, rprod, nrand * sizeof(fr_t)); nrand++;
    }
    if (!rc) { memcpy(claims_out, claims, np * sizeof(fr_t)); memcpy(rand_out, rand, nrand * sizeof(fr_t)); *nrand_out = nrand; }
    free(claims); free(coeffs); free(rand); free(rprod);
    return rc;
}

/* ------------------------------------------------------------------ sparse_mlpoly.rs: the evaluation proof */
typedef struct { fr_t init, audit; fr_t read[3], write[3]; } evals4_t;  /* (init, read_vec, write_vec, audit) */
typedef struct {
    uint8_t *comm_derefs; size_t rows_derefs;                         /* DerefsCommitment */
    /* ProductLayerProof */
    evals4_t eval_row, eval_col; fr_t dotp_left[3], dotp_right[3];
    pcbatch_t proof_mem, proof_ops;
    /* HashLayerProof */
    fr_t h_row_addr[3], h_row_read_ts[3], h_row_audit, h_col_addr[3], h_col_read_ts[3], h_col_audit, h_val[3], h_deref_row[3], h_deref_col[3];
    dplog_t pe_ops, pe_mem, pe_derefs;
} evalproof_t;
static void evalproof_free(evalproof_t *p) {
    free(p->comm_derefs); pcbatch_free(&p->proof_mem); pcbatch_free(&p->proof_ops);
    free(p->pe_ops.Lv); free(p->pe_ops.Rv); free(p->pe_mem.Lv); free(p->pe_mem.Rv); free(p->pe_derefs.Lv); free(p->pe_derefs.Rv);
}
static void hash3(fr_t *o, const fr_t *addr, const fr_t *val, const fr_t *ts, const fr_t *r_hash, const fr_t *r_hash_sqr, const fr_t *r_multiset) {
    fr_t t, u; fr_mul(&t, ts, r_hash_sqr); fr_mul(&u, val, r_hash); fr_add(&t, &t, &u); fr_add(&t, &t, addr); fr_sub(o, &t, r_multiset);   /* ts * r^2 + val * r + addr - gamma */
}
/* Layers::new / build_hash_layer for one of {row, col}: init, audit (M cells) and read, write per matrix (N ops) -> product circuits */
static void build_layers(pcirc_t *init, pcirc_t *audit, pcirc_t rd[3], pcirc_t wr[3], const fr_t *eval_table, size_t M, fr_t *const addr_f[3], fr_t *const derefs[3],
                         fr_t *const read_ts[3], const fr_t *audit_ts, size_t N, const fr_t *r_hash, const fr_t *r_multiset) {
    fr_t r2; fr_mul(&r2, r_hash, r_hash);
    fr_t *tmp = (fr_t *)malloc((M > N ? M : N) * sizeof(fr_t));
#pragma omp parallel for num_threads(g_threads) schedule(static) if (M >= 4096)
    for (size_t i = 0; i < M; i++) { fr_t a; fr_from_u64(&a, i); hash3(&tmp[i], &a, &eval_table[i], &FR_ZERO, r_hash, &r2, r_multiset); }
    pcirc_new(init, tmp, M);
#pragma omp parallel for num_threads(g_threads) schedule(static) if (M >= 4096)
    for (size_t i = 0; i < M; i++) { fr_t a; fr_from_u64(&a, i); hash3(&tmp[i], &a, &eval_table[i], &audit_ts[i], r_hash, &r2, r_multiset); }
    pcirc_new(audit, tmp, M);
    for (int k = 0; k < 3; k++) {
#pragma omp parallel for num_threads(g_threads) schedule(static) if (N >= 4096)
        for (size_t i = 0; i < N; i++) hash3(&tmp[i], &addr_f[k][i], &derefs[k][i], &read_ts[k][i], r_hash, &r2, r_multiset);
        pcirc_new(&rd[k], tmp, N);
#pragma omp parallel for num_threads(g_threads) schedule(static) if (N >= 4096)
        for (size_t i = 0; i < N; i++) { fr_t w; fr_add(&w, &read_ts[k][i], &FR_ONE); hash3(&tmp[i], &addr_f[k][i], &derefs[k][i], &w, r_hash, &r2, r_multiset); }
        pcirc_new(&wr[k], tmp, N);
    }
    free(tmp);
}
/* SparseMatPolyEvalProof::equalize: the shorter point is extended with zeros at the FRONT */
static void equalize(const fr_t *rx, size_t nrx, const fr_t *ry, size_t nry, fr_t *rxe, fr_t *rye, size_t *n) {
    size_t m = nrx > nry ? nrx : nry; *n = m;
    for (size_t i = 0; i < m - nrx; i++) rxe[i] = FR_ZERO;
    memcpy(rxe + (m - nrx), rx, nrx * sizeof(fr_t));
    for (size_t i = 0; i < m - nry; i++) rye[i] = FR_ZERO;
    memcpy(rye + (m - nry), ry, nry * sizeof(fr_t));
}
static void append_evals4(transcript_t *tr, const evals4_t *e, const char *l_init, const char *l_read, const char *l_write, const char *l_audit) {
    tr_append_scalar(tr, l_init, &e->init); tr_append_scalars(tr, l_read, e->read, 3); tr_append_scalars(tr, l_write, e->write, 3); tr_append_scalar(tr, l_audit, &e->audit);
}

/* R1CSEvalProof::prove -> SparseMatPolyEvalProof::prove */
static void evalproof_prove(evalproof_t *P, const dense_rep_t *d, const fr_t *rx, size_t nrx, const fr_t *ry, size_t nry, const fr_t evals[3], const orc_snark_gens *g,
                            transcript_t *tr, transcript_t *tape) {
    memset(P, 0, sizeof *P);
    const size_t N = d->N, M = d->M;
    tr_protocol_name(tr, "Sparse polynomial evaluation proof");
    fr_t rxe[64], rye[64]; size_t nm; equalize(rx, nrx, ry, nry, rxe, rye, &nm);
    assert(((size_t)1 << nm) == M);
    fr_t *mem_rx = (fr_t *)malloc(M * sizeof(fr_t)), *mem_ry = (fr_t *)malloc(M * sizeof(fr_t));
    orc_eq_evals(rxe, nm, mem_rx); orc_eq_evals(rye, nm, mem_ry);
    /* dense.deref(mem_rx, mem_ry): row_ops_val[k][i] = mem_rx[row_addr[k][i]], col likewise; comb = merge(row.., col..) */
    fr_t *drow[3], *dcol[3];
    for (int k = 0; k < 3; k++) {
        drow[k] = (fr_t *)malloc(N * sizeof(fr_t)); dcol[k] = (fr_t *)malloc(N * sizeof(fr_t));
        for (size_t i = 0; i < N; i++) { drow[k][i] = mem_rx[d->row_addr[k][i]]; dcol[k][i] = mem_ry[d->col_addr[k][i]]; }
    }
    size_t comb_len = next_pow2(6 * N), vars_derefs = ilog2(comb_len);
    assert(vars_derefs == g->vars_derefs);
    fr_t *comb = (fr_t *)calloc(comb_len, sizeof(fr_t));
    for (int k = 0; k < 3; k++) { memcpy(comb + (size_t)k * N, drow[k], N * sizeof(fr_t)); memcpy(comb + (size_t)(3 + k) * N, dcol[k], N * sizeof(fr_t)); }
    P->comm_derefs = poly_commit(comb, vars_derefs, &g->derefs, &P->rows_derefs);
    /* DerefsCommitment::append_to_transcript(b"comm_poly_row_col_ops_val") */
    tr_append(tr, "derefs_commitment", (const uint8_t *)"begin_derefs_commitment", 23);
    append_poly_commitment(tr, "comm_poly_row_col_ops_val", P->comm_derefs, P->rows_derefs);
    tr_append(tr, "derefs_commitment", (const uint8_t *)"end_derefs_commitment", 21);
    fr_t r_mem_check[2]; tr_challenge_vector(tr, "challenge_r_hash", r_mem_check, 2);
    /* PolyEvalNetwork::new */
    pcirc_t row_init, row_audit, row_rd[3], row_wr[3], col_init, col_audit, col_rd[3], col_wr[3];
    build_layers(&row_init, &row_audit, row_rd, row_wr, mem_rx, M, d->row_addr_f, drow, d->row_read_ts, d->row_audit_ts, N, &r_mem_check[0], &r_mem_check[1]);
    build_layers(&col_init, &col_audit, col_rd, col_wr, mem_ry, M, d->col_addr_f, dcol, d->col_read_ts, d->col_audit_ts, N, &r_mem_check[0], &r_mem_check[1]);
    /* PolyEvalNetworkProof::prove */
    tr_protocol_name(tr, "Sparse polynomial evaluation proof");
    /* ---- ProductLayerProof::prove */
    tr_protocol_name(tr, "Sparse polynomial product layer proof");
    pcirc_eval(&P->eval_row.init, &row_init); pcirc_eval(&P->eval_row.audit, &row_audit); pcirc_eval(&P->eval_col.init, &col_init); pcirc_eval(&P->eval_col.audit, &col_audit);
    for (int k = 0; k < 3; k++) { pcirc_eval(&P->eval_row.read[k], &row_rd[k]); pcirc_eval(&P->eval_row.write[k], &row_wr[k]); pcirc_eval(&P->eval_col.read[k], &col_rd[k]); pcirc_eval(&P->eval_col.write[k], &col_wr[k]); }
    append_evals4(tr, &P->eval_row, "claim_row_eval_init", "claim_row_eval_read", "claim_row_eval_write", "claim_row_eval_audit");
    append_evals4(tr, &P->eval_col, "claim_col_eval_init", "claim_col_eval_read", "claim_col_eval_write", "claim_col_eval_audit");
    /* the evaluation itself as two dot-product circuits per matrix: halves of (row_ops_val, col_ops_val, val) */
    fr_t *dl[6], *dr[6], *dw[6];
    const size_t H = N / 2;
    for (int k = 0; k < 3; k++) {
        for (int half = 0; half < 2; half++) {
            fr_t *l = (fr_t *)malloc((H ? H : 1) * sizeof(fr_t)), *r = (fr_t *)malloc((H ? H : 1) * sizeof(fr_t)), *w = (fr_t *)malloc((H ? H : 1) * sizeof(fr_t));
            memcpy(l, drow[k] + half * H, H * sizeof(fr_t)); memcpy(r, dcol[k] + half * H, H * sizeof(fr_t)); memcpy(w, d->val[k] + half * H, H * sizeof(fr_t));
            dl[2 * k + half] = l; dr[2 * k + half] = r; dw[2 * k + half] = w;
            fr_t acc = FR_ZERO, t; for (size_t x = 0; x < H; x++) { fr_mul(&t, &l[x], &r[x]); fr_mul(&t, &t, &w[x]); fr_add(&acc, &acc, &t); }
            if (half == 0) P->dotp_left[k] = acc; else P->dotp_right[k] = acc;
        }
        tr_append_scalar(tr, "claim_eval_dotp_left", &P->dotp_left[k]); tr_append_scalar(tr, "claim_eval_dotp_right", &P->dotp_right[k]);
        { fr_t s; fr_add(&s, &P->dotp_left[k], &P->dotp_right[k]); assert(fr_eq(&s, &evals[k])); (void)s; }
    }
    fr_t rand_ops[64], rand_mem[64]; size_t n_rand_ops, n_rand_mem;
    {   /* row reads A, B, C; row writes; col reads; col writes — then the six dot-product halves */
        pcirc_t ops[12] = {row_rd[0], row_rd[1], row_rd[2], row_wr[0], row_wr[1], row_wr[2], col_rd[0], col_rd[1], col_rd[2], col_wr[0], col_wr[1], col_wr[2]};
        pcbatch_prove(&P->proof_ops, ops, 12, dl, dr, dw, 6, H, tr, rand_ops, &n_rand_ops);
        pcirc_t mem[4] = {row_init, row_audit, col_init, col_audit};
        pcbatch_prove(&P->proof_mem, mem, 4, NULL, NULL, NULL, 0, 0, tr, rand_mem, &n_rand_mem);
    }
    /* ---- HashLayerProof::prove((rand_mem, rand_ops)) */
    tr_protocol_name(tr, "Sparse polynomial hash layer proof");
    for (int k = 0; k < 3; k++) { poly_evaluate(&P->h_deref_row[k], drow[k], n_rand_ops, rand_ops); poly_evaluate(&P->h_deref_col[k], dcol[k], n_rand_ops, rand_ops); }
    {   /* DerefsEvalProof::prove */
        tr_protocol_name(tr, "Derefs evaluation proof");
        fr_t ev[8], ch[3], joint, rj[64];
        for (int k = 0; k < 3; k++) { ev[k] = P->h_deref_row[k]; ev[3 + k] = P->h_deref_col[k]; } ev[6] = ev[7] = FR_ZERO;
        tr_append_scalars(tr, "evals_ops_val", ev, 8);
        tr_challenge_vector(tr, "challenge_combine_n_to_one", ch, 3);
        reduce_evals(&joint, ev, 8, ch);
        memcpy(rj, ch, 3 * sizeof(fr_t)); memcpy(rj + 3, rand_ops, n_rand_ops * sizeof(fr_t));
        tr_append_scalar(tr, "joint_claim_eval", &joint);
        polyeval_prove(&P->pe_derefs, comb, vars_derefs, rj, &joint, &g->derefs, tr, tape);
    }
    for (int k = 0; k < 3; k++) {
        poly_evaluate(&P->h_row_addr[k], d->row_addr_f[k], n_rand_ops, rand_ops); poly_evaluate(&P->h_row_read_ts[k], d->row_read_ts[k], n_rand_ops, rand_ops);
        poly_evaluate(&P->h_col_addr[k], d->col_addr_f[k], n_rand_ops, rand_ops); poly_evaluate(&P->h_col_read_ts[k], d->col_read_ts[k], n_rand_ops, rand_ops);
        poly_evaluate(&P->h_val[k], d->val[k], n_rand_ops, rand_ops);
    }
    poly_evaluate(&P->h_row_audit, d->row_audit_ts, n_rand_mem, rand_mem); poly_evaluate(&P->h_col_audit, d->col_audit_ts, n_rand_mem, rand_mem);
    {   /* one decommitment of comb_ops at rand_ops */
        fr_t ev[16], ch[4], joint, rj[64];
        for (int k = 0; k < 3; k++) { ev[k] = P->h_row_addr[k]; ev[3 + k] = P->h_row_read_ts[k]; ev[6 + k] = P->h_col_addr[k]; ev[9 + k] = P->h_col_read_ts[k]; ev[12 + k] = P->h_val[k]; }
        ev[15] = FR_ZERO;
        tr_append_scalars(tr, "claim_evals_ops", ev, 16);
        tr_challenge_vector(tr, "challenge_combine_n_to_one", ch, 4);
        reduce_evals(&joint, ev, 16, ch);
        memcpy(rj, ch, 4 * sizeof(fr_t)); memcpy(rj + 4, rand_ops, n_rand_ops * sizeof(fr_t));
        tr_append_scalar(tr, "joint_claim_eval_ops", &joint);
        polyeval_prove(&P->pe_ops, d->comb_ops, d->vars_comb_ops, rj, &joint, &g->ops, tr, tape);
    }
    {   /* one decommitment of comb_mem at rand_mem */
        fr_t ev[2] = {P->h_row_audit, P->h_col_audit}, ch[1], joint, rj[64];
        tr_append_scalars(tr, "claim_evals_mem", ev, 2);
        tr_challenge_vector(tr, "challenge_combine_two_to_one", ch, 1);
        reduce_evals(&joint, ev, 2, ch);
        rj[0] = ch[0]; memcpy(rj + 1, rand_mem, n_rand_mem * sizeof(fr_t));
        tr_append_scalar(tr, "joint_claim_eval_mem", &joint);
        polyeval_prove(&P->pe_mem, d->comb_mem, d->vars_comb_mem, rj, &joint, &g->mem, tr, tape);
    }
    for (int k = 0; k < 6; k++) { free(dl[k]); free(dr[k]); free(dw[k]); }
    for (int k = 0; k < 3; k++) { free(drow[k]); free(dcol[k]); pcirc_free(&row_rd[k]); pcirc_free(&row_wr[k]); pcirc_free(&col_rd[k]); pcirc_free(&col_wr[k]); }
    pcirc_free(&row_init); pcirc_free(&row_audit); pcirc_free(&col_init); pcirc_free(&col_audit);
    free(comb); free(mem_rx); free(mem_ry);
}

/* HashLayerProof::verify_helper */
static int hash_verify_helper(const fr_t *rand_mem, size_t nm, const evals4_t *claims, const fr_t ops_val[3], const fr_t ops_addr[3], const fr_t read_ts[3], const fr_t *audit_ts,
                              const fr_t *r, const fr_t *r_hash, const fr_t *r_multiset) {
    fr_t r2, init_addr = FR_ZERO, init_val = FR_ONE, t, u, h;
    fr_mul(&r2, r_hash, r_hash);
    for (size_t i = 0; i < nm; i++) {                                 /* IdentityPolynomial::evaluate and EqPolynomial(r)::evaluate at rand_mem */
        fr_t p2, om1, om2; fr_from_u64(&p2, (uint64_t)1 << (nm - i - 1)); fr_mul(&t, &p2, &rand_mem[i]); fr_add(&init_addr, &init_addr, &t);
        fr_mul(&t, &r[i], &rand_mem[i]); fr_sub(&om1, &FR_ONE, &r[i]); fr_sub(&om2, &FR_ONE, &rand_mem[i]); fr_mul(&u, &om1, &om2); fr_add(&t, &t, &u); fr_mul(&init_val, &init_val, &t);
    }
    hash3(&h, &init_addr, &init_val, &FR_ZERO, r_hash, &r2, r_multiset);
    if (!fr_eq(&h, &claims->init)) return ORC_ERR_VERIFY_INTERNAL;
    for (int k = 0; k < 3; k++) {
        hash3(&h, &ops_addr[k], &ops_val[k], &read_ts[k], r_hash, &r2, r_multiset);
        if (!fr_eq(&h, &claims->read[k])) return ORC_ERR_VERIFY_INTERNAL;
        fr_t w; fr_add(&w, &read_ts[k], &FR_ONE);
        hash3(&h, &ops_addr[k], &ops_val[k], &w, r_hash, &r2, r_multiset);
        if (!fr_eq(&h, &claims->write[k])) return ORC_ERR_VERIFY_INTERNAL;
    }
    hash3(&h, &init_addr, &init_val, audit_ts, r_hash, &r2, r_multiset);
    return fr_eq(&h, &claims->audit) ? ORC_OK : ORC_ERR_VERIFY_INTERNAL;
}
/* R1CSEvalProof::verify -> SparseMatPolyEvalProof::verify -> PolyEvalNetworkProof::verify */
static int evalproof_verify(const evalproof_t *P, const orc_snark_comm *c, const fr_t *rx, size_t nrx, const fr_t *ry, size_t nry, const fr_t evals[3], const orc_snark_gens *g, transcript_t *tr) {
    int rc;
    tr_protocol_name(tr, "Sparse polynomial evaluation proof");
    fr_t rxe[64], rye[64]; size_t nm; equalize(rx, nrx, ry, nry, rxe, rye, &nm);
    const size_t N = c->num_ops, M = c->num_mem_cells;
    if (((size_t)1 << nm) != M || N != next_pow2(N) || N < 2) return ORC_ERR_VERIFY_INTERNAL;
    tr_append(tr, "derefs_commitment", (const uint8_t *)"begin_derefs_commitment", 23);
    append_poly_commitment(tr, "comm_poly_row_col_ops_val", P->comm_derefs, P->rows_derefs);
    tr_append(tr, "derefs_commitment", (const uint8_t *)"end_derefs_commitment", 21);
    fr_t r_mem_check[2]; tr_challenge_vector(tr, "challenge_r_hash", r_mem_check, 2);
    tr_protocol_name(tr, "Sparse polynomial evaluation proof");
    /* ---- ProductLayerProof::verify */
    tr_protocol_name(tr, "Sparse polynomial product layer proof");
    for (int side = 0; side < 2; side++) {                            /* subset check: init * prod(writes) == prod(reads) * audit */
        const evals4_t *e = side ? &P->eval_col : &P->eval_row;
        fr_t ws = FR_ONE, rs = FR_ONE, l, r;
        for (int k = 0; k < 3; k++) { fr_mul(&ws, &ws, &e->write[k]); fr_mul(&rs, &rs, &e->read[k]); }
        fr_mul(&l, &e->init, &ws); fr_mul(&r, &rs, &e->audit);
        if (!fr_eq(&l, &r)) return ORC_ERR_VERIFY_INTERNAL;
        if (side == 0) append_evals4(tr, e, "claim_row_eval_init", "claim_row_eval_read", "claim_row_eval_write", "claim_row_eval_audit");
        else append_evals4(tr, e, "claim_col_eval_init", "claim_col_eval_read", "claim_col_eval_write", "claim_col_eval_audit");
    }
    fr_t claims_dotp_circuit[6];
    for (int k = 0; k < 3; k++) {
        fr_t s; fr_add(&s, &P->dotp_left[k], &P->dotp_right[k]);
        if (!fr_eq(&s, &evals[k])) return ORC_ERR_VERIFY_INTERNAL;
        tr_append_scalar(tr, "claim_eval_dotp_left", &P->dotp_left[k]); tr_append_scalar(tr, "claim_eval_dotp_right", &P->dotp_right[k]);
        claims_dotp_circuit[2 * k] = P->dotp_left[k]; claims_dotp_circuit[2 * k + 1] = P->dotp_right[k];
    }
    fr_t claims_prod[12], claims_ops[12], claims_dotp[9], rand_ops[64], claims_mem_in[4], claims_mem[4], rand_mem[64], dummy[3]; size_t n_rand_ops, n_rand_mem;
    for (int k = 0; k < 3; k++) { claims_prod[k] = P->eval_row.read[k]; claims_prod[3 + k] = P->eval_row.write[k]; claims_prod[6 + k] = P->eval_col.read[k]; claims_prod[9 + k] = P->eval_col.write[k]; }
    if ((rc = pcbatch_verify(&P->proof_ops, claims_prod, 12, claims_dotp_circuit, 6, N, tr, claims_ops, claims_dotp, rand_ops, &n_rand_ops))) return rc;
    claims_mem_in[0] = P->eval_row.init; claims_mem_in[1] = P->eval_row.audit; claims_mem_in[2] = P->eval_col.init; claims_mem_in[3] = P->eval_col.audit;
    if ((rc = pcbatch_verify(&P->proof_mem, claims_mem_in, 4, NULL, 0, M, tr, claims_mem, dummy, rand_mem, &n_rand_mem))) return rc;
    /* ---- HashLayerProof::verify */
    tr_protocol_name(tr, "Sparse polynomial hash layer proof");
    {   /* DerefsEvalProof::verify */
        tr_protocol_name(tr, "Derefs evaluation proof");
        fr_t ev[8], ch[3], joint, rj[64];
        for (int k = 0; k < 3; k++) { ev[k] = P->h_deref_row[k]; ev[3 + k] = P->h_deref_col[k]; } ev[6] = ev[7] = FR_ZERO;
        tr_append_scalars(tr, "evals_ops_val", ev, 8);
        tr_challenge_vector(tr, "challenge_combine_n_to_one", ch, 3);
        reduce_evals(&joint, ev, 8, ch);
        memcpy(rj, ch, 3 * sizeof(fr_t)); memcpy(rj + 3, rand_ops, n_rand_ops * sizeof(fr_t));
        tr_append_scalar(tr, "joint_claim_eval", &joint);
        if ((rc = polyeval_verify_plain(&P->pe_derefs, g->vars_derefs, rj, &joint, P->comm_derefs, P->rows_derefs, &g->derefs, tr))) return rc;
    }
    for (int k = 0; k < 3; k++)                                       /* the dot-product claims must be the decommitted values */
        if (!fr_eq(&claims_dotp[3 * k], &P->h_deref_row[k]) || !fr_eq(&claims_dotp[3 * k + 1], &P->h_deref_col[k]) || !fr_eq(&claims_dotp[3 * k + 2], &P->h_val[k])) return ORC_ERR_VERIFY_INTERNAL;
    {
        fr_t ev[16], ch[4], joint, rj[64];
        for (int k = 0; k < 3; k++) { ev[k] = P->h_row_addr[k]; ev[3 + k] = P->h_row_read_ts[k]; ev[6 + k] = P->h_col_addr[k]; ev[9 + k] = P->h_col_read_ts[k]; ev[12 + k] = P->h_val[k]; }
        ev[15] = FR_ZERO;
        tr_append_scalars(tr, "claim_evals_ops", ev, 16);
        tr_challenge_vector(tr, "challenge_combine_n_to_one", ch, 4);
        reduce_evals(&joint, ev, 16, ch);
        memcpy(rj, ch, 4 * sizeof(fr_t)); memcpy(rj + 4, rand_ops, n_rand_ops * sizeof(fr_t));
        tr_append_scalar(tr, "joint_claim_eval_ops", &joint);
        if ((rc = polyeval_verify_plain(&P->pe_ops, g->vars_ops, rj, &joint, c->comm_ops, c->rows_ops, &g->ops, tr))) return rc;
    }
    {
        fr_t ev[2] = {P->h_row_audit, P->h_col_audit}, ch[1], joint, rj[64];
        tr_append_scalars(tr, "claim_evals_mem", ev, 2);
        tr_challenge_vector(tr, "challenge_combine_two_to_one", ch, 1);
        reduce_evals(&joint, ev, 2, ch);
        rj[0] = ch[0]; memcpy(rj + 1, rand_mem, n_rand_mem * sizeof(fr_t));
        tr_append_scalar(tr, "joint_claim_eval_mem", &joint);
        if ((rc = polyeval_verify_plain(&P->pe_mem, g->vars_mem, rj, &joint, c->comm_mem, c->rows_mem, &g->mem, tr))) return rc;
    }
    evals4_t crow, ccol;                                              /* the product layer's claims at (rand_mem, rand_ops) */
    crow.init = claims_mem[0]; crow.audit = claims_mem[1]; ccol.init = claims_mem[2]; ccol.audit = claims_mem[3];
    for (int k = 0; k < 3; k++) { crow.read[k] = claims_ops[k]; crow.write[k] = claims_ops[3 + k]; ccol.read[k] = claims_ops[6 + k]; ccol.write[k] = claims_ops[9 + k]; }
    if (n_rand_mem != nm) return ORC_ERR_VERIFY_INTERNAL;
    if ((rc = hash_verify_helper(rand_mem, nm, &crow, P->h_deref_row, P->h_row_addr, P->h_row_read_ts, &P->h_row_audit, rxe, &r_mem_check[0], &r_mem_check[1]))) return rc;
    return hash_verify_helper(rand_mem, nm, &ccol, P->h_deref_col, P->h_col_addr, P->h_col_read_ts, &P->h_col_audit, rye, &r_mem_check[0], &r_mem_check[1]);
}

/* ------------------------------------------------------------------ bincode of R1CSEvalProof */
static void wb_frs(wbuf_t *w, const fr_t *v, size_t n) { wb_u64(w, n); for (size_t i = 0; i < n; i++) wb_fr(w, &v[i]); }
static void wb_evals4(wbuf_t *w, const evals4_t *e) { wb_fr(w, &e->init); wb_frs(w, e->read, 3); wb_frs(w, e->write, 3); wb_fr(w, &e->audit); }
static void wb_pcbatch(wbuf_t *w, const pcbatch_t *p) {
    wb_u64(w, p->nlayers);
    for (size_t i = 0; i < p->nlayers; i++) {
        const layerproof_t *L = &p->layers[i];
        wb_u64(w, L->sc.rounds); for (size_t j = 0; j < L->sc.rounds; j++) wb_frs(w, &L->sc.coeffs[3 * j], 3);
        wb_frs(w, L->left, L->np); wb_frs(w, L->right, L->np);
    }
    wb_frs(w, p->dl, p->nd); wb_frs(w, p->dr, p->nd); wb_frs(w, p->dw, p->nd);
}
static void wb_dplog(wbuf_t *w, const dplog_t *d) {
    wb_u64(w, d->n); wb(w, d->Lv, 32 * d->n); wb_u64(w, d->n); wb(w, d->Rv, 32 * d->n);
    wb(w, d->delta, 32); wb(w, d->beta, 32); wb_fr(w, &d->z1); wb_fr(w, &d->z2);
}
static void evalproof_serialize(wbuf_t *w, const evalproof_t *P) {
    wb_u64(w, P->rows_derefs); wb(w, P->comm_derefs, 32 * P->rows_derefs);
    wb_evals4(w, &P->eval_row); wb_evals4(w, &P->eval_col); wb_frs(w, P->dotp_left, 3); wb_frs(w, P->dotp_right, 3);
    wb_pcbatch(w, &P->proof_mem); wb_pcbatch(w, &P->proof_ops);
    wb_frs(w, P->h_row_addr, 3); wb_frs(w, P->h_row_read_ts, 3); wb_fr(w, &P->h_row_audit);
    wb_frs(w, P->h_col_addr, 3); wb_frs(w, P->h_col_read_ts, 3); wb_fr(w, &P->h_col_audit);
    wb_frs(w, P->h_val, 3); wb_frs(w, P->h_deref_row, 3); wb_frs(w, P->h_deref_col, 3);
    wb_dplog(w, &P->pe_ops); wb_dplog(w, &P->pe_mem); wb_dplog(w, &P->pe_derefs);
}
static void rb_frs_fixed(rbuf_t *r, fr_t *v, size_t n) { if (rb_u64(r) != n) r->bad = 1; for (size_t i = 0; i < n && !r->bad; i++) rb_fr(r, &v[i]); }
static fr_t *rb_frs(rbuf_t *r, size_t *n, size_t max) {
    uint64_t k = rb_u64(r); if (r->bad || k > max) { r->bad = 1; k = 0; }
    fr_t *v = (fr_t *)calloc(k ? k : 1, sizeof(fr_t)); for (size_t i = 0; i < k && !r->bad; i++) rb_fr(r, &v[i]); *n = (size_t)k; return v;
}
static void rb_evals4(rbuf_t *r, evals4_t *e) { rb_fr(r, &e->init); rb_frs_fixed(r, e->read, 3); rb_frs_fixed(r, e->write, 3); rb_fr(r, &e->audit); }
static void rb_pcbatch(rbuf_t *r, pcbatch_t *p) {
    uint64_t nl = rb_u64(r); if (r->bad || nl > 64) { r->bad = 1; nl = 0; }
    p->nlayers = nl; p->layers = (layerproof_t *)calloc(nl ? nl : 1, sizeof(layerproof_t));
    for (size_t i = 0; i < nl && !r->bad; i++) {
        layerproof_t *L = &p->layers[i];
        uint64_t rounds = rb_u64(r); if (r->bad || rounds > 64) { r->bad = 1; break; }
        L->sc.rounds = rounds; L->sc.coeffs = (fr_t *)calloc(3 * rounds + 1, sizeof(fr_t));
        for (size_t j = 0; j < rounds && !r->bad; j++) rb_frs_fixed(r, &L->sc.coeffs[3 * j], 3);
        size_t a, b; L->left = rb_frs(r, &a, 64); L->right = rb_frs(r, &b, 64); if (a != b) r->bad = 1; L->np = a;
    }
    size_t a, b, c; p->dl = rb_frs(r, &a, 64); p->dr = rb_frs(r, &b, 64); p->dw = rb_frs(r, &c, 64); if (a != b || b != c) r->bad = 1; p->nd = a;
}
static void rb_dplog(rbuf_t *r, dplog_t *d) {
    size_t nl, nr; d->Lv = rb_vec32(r, &nl, 64); d->Rv = rb_vec32(r, &nr, 64); if (nl != nr) r->bad = 1; d->n = nl;
    rb(r, d->delta, 32); rb(r, d->beta, 32); rb_fr(r, &d->z1); rb_fr(r, &d->z2);
}
static void evalproof_parse(rbuf_t *r, evalproof_t *P) {
    memset(P, 0, sizeof *P);
    P->comm_derefs = rb_vec32(r, &P->rows_derefs, (size_t)1 << 24);
    rb_evals4(r, &P->eval_row); rb_evals4(r, &P->eval_col); rb_frs_fixed(r, P->dotp_left, 3); rb_frs_fixed(r, P->dotp_right, 3);
    rb_pcbatch(r, &P->proof_mem); rb_pcbatch(r, &P->proof_ops);
    rb_frs_fixed(r, P->h_row_addr, 3); rb_frs_fixed(r, P->h_row_read_ts, 3); rb_fr(r, &P->h_row_audit);
    rb_frs_fixed(r, P->h_col_addr, 3); rb_frs_fixed(r, P->h_col_read_ts, 3); rb_fr(r, &P->h_col_audit);
    rb_frs_fixed(r, P->h_val, 3); rb_frs_fixed(r, P->h_deref_row, 3); rb_frs_fixed(r, P->h_deref_col, 3);
    rb_dplog(r, &P->pe_ops); rb_dplog(r, &P->pe_mem); rb_dplog(r, &P->pe_derefs);
}

/* ------------------------------------------------------------------ lib.rs SNARK::prove / SNARK::verify */
/* stage_ms: the seven R1CSProof stages of orc_nizk_prove, then [7] = R1CSEvalProof, [8] = total */
int orc_snark_prove(const orc_instance *I, const orc_snark_comm *comm, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ni, const orc_snark_gens *g,
                    const uint8_t *tlabel, size_t tlabel_len, const uint8_t seed32[32], uint8_t **proof, size_t *proof_len, double *ms) {
    size_t N = I->num_cons, V = I->num_vars;
    if (nvars > V) return ORC_ERR_INVALID_NUM_VARS;
    if (ni != I->num_inputs) return ORC_ERR_INVALID_NUM_INPUTS;
    if (!comm->d.comb_ops) return ORC_ERR_VERIFY_INTERNAL;            /* a parsed commitment carries no decommitment */
    fr_t *vars = (fr_t *)calloc(V, sizeof(fr_t)), *inputs = (fr_t *)calloc(ni + 1, sizeof(fr_t));
    for (size_t i = 0; i < nvars; i++) if (!fr_from_bytes(&vars[i], vars32 + 32 * i)) { free(vars); free(inputs); return ORC_ERR_INVALID_SCALAR; }
    for (size_t i = 0; i < ni; i++) if (!fr_from_bytes(&inputs[i], inputs32 + 32 * i)) { free(vars); free(inputs); return ORC_ERR_INVALID_SCALAR; }
    double t_start = now_ms(), st[9] = {0};
    transcript_t tr, tape; nizk_t P; memset(&P, 0, sizeof P);
    tr_init(&tr, (const char *)tlabel, tlabel_len);
    tape_init(&tape, seed32);
    tr_protocol_name(&tr, "Spartan SNARK proof");
    append_comm(&tr, comm);                                           /* comm.comm.append_to_transcript(b"comm", ..) */
    r1cs_prove(I, vars, inputs, ni, g->sat, &tr, &tape, &P, st);
    size_t nrx = ilog2(N), nry = ilog2(2 * V);
    fr_t inst_evals[3];
    inst_evaluate(I, P.rx, nrx, P.ry, nry, inst_evals);
    tr_append_scalar(&tr, "Ar_claim", &inst_evals[0]); tr_append_scalar(&tr, "Br_claim", &inst_evals[1]); tr_append_scalar(&tr, "Cr_claim", &inst_evals[2]);
    double t0 = now_ms();
    evalproof_t E;
    evalproof_prove(&E, &comm->d, P.rx, nrx, P.ry, nry, inst_evals, g, &tr, &tape);
    st[7] = now_ms() - t0;
    wbuf_t w = {0, 0, 0};
    r1cs_serialize_body(&w, &P);
    for (int k = 0; k < 3; k++) wb_fr(&w, &inst_evals[k]);
    evalproof_serialize(&w, &E);
    *proof = w.p; *proof_len = w.len;
    st[8] = now_ms() - t_start;
    if (ms) memcpy(ms, st, sizeof st);
    evalproof_free(&E); nizk_free(&P); free(vars); free(inputs);
    return ORC_OK;
}
int orc_snark_verify(const orc_snark_comm *comm, const uint8_t *inputs32, size_t ni, const orc_snark_gens *g, const uint8_t *tlabel, size_t tlabel_len,
                     const uint8_t *proof, size_t proof_len) {
    size_t N = comm->num_cons, V = comm->num_vars, nrx = ilog2(N), nry = ilog2(2 * V);
    if (ni != comm->num_inputs) return ORC_ERR_INVALID_NUM_INPUTS;
    nizk_t P; memset(&P, 0, sizeof P); evalproof_t E; fr_t inst_evals[3];
    rbuf_t r = {proof, proof_len, 0, 0};
    r1cs_parse_body(&r, &P);
    for (int k = 0; k < 3; k++) rb_fr(&r, &inst_evals[k]);
    evalproof_parse(&r, &E);
    int rc = (r.bad || r.pos != r.len) ? ORC_ERR_MALFORMED_PROOF : ORC_OK;
    fr_t *inputs = (fr_t *)calloc(ni + 1, sizeof(fr_t)), *rx = (fr_t *)malloc((nrx + 1) * sizeof(fr_t)), *ry = (fr_t *)malloc((nry + 1) * sizeof(fr_t));
    for (size_t i = 0; i < ni && !rc; i++) if (!fr_from_bytes(&inputs[i], inputs32 + 32 * i)) rc = ORC_ERR_INVALID_SCALAR;
    if (!rc) {
        transcript_t tr; tr_init(&tr, (const char *)tlabel, tlabel_len);
        tr_protocol_name(&tr, "Spartan SNARK proof");
        append_comm(&tr, comm);
        rc = r1cs_verify(&P, N, V, inputs, ni, inst_evals, g->sat, &tr, rx, ry);
        if (!rc) {
            tr_append_scalar(&tr, "Ar_claim", &inst_evals[0]); tr_append_scalar(&tr, "Br_claim", &inst_evals[1]); tr_append_scalar(&tr, "Cr_claim", &inst_evals[2]);
            rc = evalproof_verify(&E, comm, rx, nrx, ry, nry, inst_evals, g, &tr);
        }
    }
    evalproof_free(&E); nizk_free(&P); free(inputs); free(rx); free(ry);
    return rc;
}
