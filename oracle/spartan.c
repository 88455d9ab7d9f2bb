/*
 * ORACLE — test infrastructure only (see spartan.h).  PARITY UNPINNED against the (absent) reference source.
 * Restates upstream libspartan, file by file [RECALL, SURVEY.md App. A]:
 *   commitments.rs, dense_mlpoly.rs, sparse_mlpoly.rs, r1csinstance.rs, unipoly.rs, sumcheck.rs, nizk/mod.rs,
 *   nizk/bullet.rs, r1csproof.rs, lib.rs, random.rs.
 * Deviation kept deliberately small and documented in DESIGN.md: RandomTape takes a caller seed instead of OsRng.
 */
#include "spartan.h"
#include "internal.h"
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }
double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

size_t ilog2(size_t n) { size_t l = 0; while (((size_t)1 << l) < n) l++; return l; }
size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

void orc_fr_from_canon(fr_t *o, const uint8_t *b, size_t n) { for (size_t i = 0; i < n; i++) { int ok = fr_from_bytes(&o[i], b + 32 * i); assert(ok); (void)ok; } }
void orc_fr_to_canon(uint8_t *b, const fr_t *a, size_t n) { for (size_t i = 0; i < n; i++) fr_to_bytes(b + 32 * i, &a[i]); }
void orc_buf_free(void *p) { free(p); }

/* ------------------------------------------------------------------ commitments.rs */
/* MultiCommitGens::new: SHAKE256(label || compress(B)) XOF, n+1 chunks of 64 B -> one-way map; first n are G, last is h */
void gens_stream(ge_t *out, size_t count, const char *label) {
    shake256_t sh; shake256_init(&sh);
    shake256_absorb(&sh, (const uint8_t *)label, strlen(label));
    shake256_absorb(&sh, RISTRETTO_BASEPOINT_COMPRESSED, 32);
    for (size_t i = 0; i < count; i++) { uint8_t u[64]; shake256_squeeze(&sh, u, 64); ge_from_uniform_bytes(&out[i], u); }
}
void mcgens_from(orc_mcgens *g, const ge_t *P, size_t n, const ge_t *h) {
    g->n = n; g->G = (ge_t *)malloc(n * sizeof(ge_t)); memcpy(g->G, P, n * sizeof(ge_t)); g->h = *h;
}
void mcgens_free(orc_mcgens *g) { free(g->G); g->G = NULL; }

/* Commitments for [Scalar]: MSM(v, G) + blind*h */
void commit_vec(ge_t *o, const fr_t *v, size_t n, const fr_t *blind, const orc_mcgens *g) {
    assert(n == g->n);
    fr_t *s = (fr_t *)malloc((n + 1) * sizeof(fr_t)); ge_t *P = (ge_t *)malloc((n + 1) * sizeof(ge_t));
    memcpy(s, v, n * sizeof(fr_t)); s[n] = *blind; memcpy(P, g->G, n * sizeof(ge_t)); P[n] = g->h;
    ge_msm(o, s, P, n + 1);
    free(s); free(P);
}
void commit_scalar(ge_t *o, const fr_t *x, const fr_t *blind, const orc_mcgens *g) { assert(g->n == 1); commit_vec(o, x, 1, blind, g); }
void commit_vec_c(uint8_t out[32], const fr_t *v, size_t n, const fr_t *blind, const orc_mcgens *g) { ge_t p; commit_vec(&p, v, n, blind, g); ge_encode(out, &p); }
void commit_scalar_c(uint8_t out[32], const fr_t *x, const fr_t *blind, const orc_mcgens *g) { ge_t p; commit_scalar(&p, x, blind, g); ge_encode(out, &p); }

/* ------------------------------------------------------------------ lib.rs NIZKGens / r1csproof.rs R1CSGens */
orc_gens *orc_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs) {
    (void)num_cons;
    size_t nvp = num_vars > num_inputs + 1 ? num_vars : num_inputs + 1;
    nvp = next_pow2(nvp);
    size_t ell = ilog2(nvp);
    size_t R = (size_t)1 << (ell - ell / 2);                         /* EqPolynomial::compute_factored_lens: right = ell - ell/2 */
    size_t count = R + 2 > 5 ? R + 2 : 5;
    ge_t *P = (ge_t *)malloc(count * sizeof(ge_t));
    gens_stream(P, count, "gens_r1cs_sat");
    orc_gens *g = (orc_gens *)calloc(1, sizeof *g);
    /* DotProductProofGens::new(R,label) = MultiCommitGens::new(R+1,label).split_at(R): h is stream point R+1 */
    mcgens_from(&g->pc_n, P, R, &P[R + 1]);
    mcgens_from(&g->pc_1, &P[R], 1, &P[R + 1]);
    mcgens_from(&g->sc_1, &P[R], 1, &P[R + 1]);                      /* R1CSSumcheckGens: gens_1 = clone of gens_pc.gens.gens_1 */
    mcgens_from(&g->sc_3, P, 3, &P[3]);                              /* MultiCommitGens::new(3,label) */
    mcgens_from(&g->sc_4, P, 4, &P[4]);                              /* MultiCommitGens::new(4,label) */
    free(P);
    return g;
}
void orc_gens_free(orc_gens *g) { if (!g) return; mcgens_free(&g->pc_n); mcgens_free(&g->pc_1); mcgens_free(&g->sc_1); mcgens_free(&g->sc_3); mcgens_free(&g->sc_4); free(g); }
void orc_gens_points(const orc_gens *g, uint8_t *out) {
    for (size_t i = 0; i < g->pc_n.n; i++) ge_encode(out + 32 * i, &g->pc_n.G[i]);
    ge_encode(out + 32 * g->pc_n.n, &g->pc_1.G[0]);
    ge_encode(out + 32 * (g->pc_n.n + 1), &g->pc_n.h);
}

/* ------------------------------------------------------------------ lib.rs Instance::new */
static int build_matrix(orc_sparse *m, const orc_entry *t, size_t n, size_t num_cons, size_t num_vars, size_t num_inputs,
                        size_t nvp, size_t ncp) {
    size_t extra = (num_cons == 0 || num_cons == 1) ? (ncp > n ? ncp - n : 0) : 0;
    m->M = (orc_mentry *)malloc((n + extra + 1) * sizeof(orc_mentry)); m->n = 0;
    for (size_t i = 0; i < n; i++) {
        if (t[i].row >= num_cons) return ORC_ERR_INVALID_INDEX;
        if (t[i].col >= num_vars + 1 + num_inputs) return ORC_ERR_INVALID_INDEX;
        fr_t v; if (!fr_from_bytes(&v, t[i].val)) return ORC_ERR_INVALID_SCALAR;
        size_t col = t[i].col >= num_vars ? t[i].col + nvp - num_vars : t[i].col;
        m->M[m->n].row = t[i].row; m->M[m->n].col = col; m->M[m->n].val = v; m->n++;
    }
    /* upstream pads with explicit zero entries when num_cons is 0 or 1 */
    for (size_t i = n; i < n + extra; i++) { m->M[m->n].row = i; m->M[m->n].col = num_vars; m->M[m->n].val = FR_ZERO; m->n++; }
    return ORC_OK;
}

int orc_instance_new(size_t num_cons, size_t num_vars, size_t num_inputs, const orc_entry *A, size_t nA,
                     const orc_entry *B, size_t nB, const orc_entry *C, size_t nC, orc_instance **out) {
    size_t nvp = num_vars > num_inputs + 1 ? num_vars : num_inputs + 1; nvp = next_pow2(nvp);
    size_t ncp = num_cons < 2 ? 2 : next_pow2(num_cons);
    orc_instance *I = (orc_instance *)calloc(1, sizeof *I);
    I->num_cons = ncp; I->num_vars = nvp; I->num_inputs = num_inputs;
    int rc;
    if ((rc = build_matrix(&I->A, A, nA, num_cons, num_vars, num_inputs, nvp, ncp)) ||
        (rc = build_matrix(&I->B, B, nB, num_cons, num_vars, num_inputs, nvp, ncp)) ||
        (rc = build_matrix(&I->C, C, nC, num_cons, num_vars, num_inputs, nvp, ncp))) { orc_instance_free(I); return rc; }
    *out = I; return ORC_OK;
}
void orc_instance_free(orc_instance *I) { if (!I) return; free(I->A.M); free(I->B.M); free(I->C.M); free(I); }

/* ------------------------------------------------------------------ dense_mlpoly.rs */
/* EqPolynomial::evals — doubling table, MSB-first index order */
void orc_eq_evals(const fr_t *r, size_t ell, fr_t *ev) {
    size_t n = (size_t)1 << ell;
    ev[0] = FR_ONE;
    if (g_threads <= 1 || n < 4096) {
        size_t size = 1;
        for (size_t j = 0; j < ell; j++) {
            size *= 2;
            for (size_t i = size - 1;; i -= 2) {       /* (0..size).rev().step_by(2) */
                fr_t s = ev[i / 2];
                fr_mul(&ev[i], &s, &r[j]);
                fr_sub(&ev[i - 1], &s, &ev[i]);
                if (i == 1) break;
            }
        }
        return;
    }
    /* same table, built level by level out of place so the level loop can be split across threads */
    fr_t *tmp = (fr_t *)malloc(n * sizeof(fr_t)), *cur = ev, *nxt = tmp;
    size_t size = 1;
    for (size_t j = 0; j < ell; j++) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (size_t k = 0; k < size; k++) { fr_mul(&nxt[2 * k + 1], &cur[k], &r[j]); fr_sub(&nxt[2 * k], &cur[k], &nxt[2 * k + 1]); }
        fr_t *t = cur; cur = nxt; nxt = t; size *= 2;
    }
    if (cur != ev) memcpy(ev, cur, n * sizeof(fr_t));
    free(tmp);
}

void orc_fold_top(fr_t *Z, size_t len, const fr_t *r) {
    size_t n = len / 2;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (n >= 4096)
    for (size_t i = 0; i < n; i++) { fr_t d; fr_sub(&d, &Z[i + n], &Z[i]); fr_mul(&d, &d, r); fr_add(&Z[i], &Z[i], &d); }
}
void orc_fold_bot(fr_t *Z, size_t len, const fr_t *r) {
    size_t n = len / 2;
    for (size_t i = 0; i < n; i++) { fr_t d; fr_sub(&d, &Z[2 * i + 1], &Z[2 * i]); fr_mul(&d, &d, r); fr_add(&Z[i], &Z[2 * i], &d); }
}

void dot(fr_t *o, const fr_t *a, const fr_t *b, size_t n) {
    fr_t acc = FR_ZERO;
    if (g_threads > 1 && n >= 4096) {
#pragma omp parallel num_threads(g_threads)
        {
            fr_t loc = FR_ZERO;
#pragma omp for schedule(static) nowait
            for (size_t i = 0; i < n; i++) { fr_t t; fr_mul(&t, &a[i], &b[i]); fr_add(&loc, &loc, &t); }
#pragma omp critical
            fr_add(&acc, &acc, &loc);
        }
    } else for (size_t i = 0; i < n; i++) { fr_t t; fr_mul(&t, &a[i], &b[i]); fr_add(&acc, &acc, &t); }
    *o = acc;
}

/* DensePolynomial::bound: LZ[i] = sum_j L[j] * Z[j*R + i] */
void orc_poly_bound(const fr_t *Z, size_t L, size_t R, const fr_t *Lv, fr_t *out) {
#pragma omp parallel for num_threads(g_threads) schedule(static) if (L * R >= 4096)
    for (size_t i = 0; i < R; i++) {
        fr_t acc = FR_ZERO;
        for (size_t j = 0; j < L; j++) { fr_t t; fr_mul(&t, &Lv[j], &Z[j * R + i]); fr_add(&acc, &acc, &t); }
        out[i] = acc;
    }
}

/* DensePolynomial::commit_inner: row i -> compress(commit(Z[R*i..R*(i+1)], blinds[i])) */
void orc_commit_rows(const fr_t *Z, size_t L, size_t R, const fr_t *blinds, const orc_mcgens *g, uint8_t *out) {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
    for (size_t i = 0; i < L; i++) commit_vec_c(out + 32 * i, Z + R * i, R, &blinds[i], g);
}

/* ------------------------------------------------------------------ sparse_mlpoly.rs / r1csinstance.rs */
static void sparse_mulvec(const orc_sparse *m, const fr_t *z, fr_t *out, size_t rows) {
    memset(out, 0, rows * sizeof(fr_t));
    for (size_t i = 0; i < m->n; i++) { fr_t t; fr_mul(&t, &m->M[i].val, &z[m->M[i].col]); fr_add(&out[m->M[i].row], &out[m->M[i].row], &t); }
}
void orc_multiply_vec(const orc_instance *I, const fr_t *z, fr_t *Az, fr_t *Bz, fr_t *Cz) {
    const orc_sparse *m[3] = {&I->A, &I->B, &I->C}; fr_t *o[3] = {Az, Bz, Cz};
#pragma omp parallel for num_threads(g_threads < 3 ? g_threads : 3) schedule(static, 1)
    for (int k = 0; k < 3; k++) sparse_mulvec(m[k], z, o[k], I->num_cons);
}
static void sparse_evaltable(const orc_sparse *m, const fr_t *rx, fr_t *out, size_t cols) {
    memset(out, 0, cols * sizeof(fr_t));
    for (size_t i = 0; i < m->n; i++) { fr_t t; fr_mul(&t, &rx[m->M[i].row], &m->M[i].val); fr_add(&out[m->M[i].col], &out[m->M[i].col], &t); }
}
void orc_eval_table_sparse(const orc_instance *I, const fr_t *rx, fr_t *eA, fr_t *eB, fr_t *eC) {
    const orc_sparse *m[3] = {&I->A, &I->B, &I->C}; fr_t *o[3] = {eA, eB, eC};
#pragma omp parallel for num_threads(g_threads < 3 ? g_threads : 3) schedule(static, 1)
    for (int k = 0; k < 3; k++) sparse_evaltable(m[k], rx, o[k], 2 * I->num_vars);
}
/* R1CSInstance::evaluate -> SparseMatPolynomial::multi_evaluate */
void inst_evaluate(const orc_instance *I, const fr_t *rx, size_t nrx, const fr_t *ry, size_t nry, fr_t ev[3]) {
    fr_t *ex = (fr_t *)malloc(((size_t)1 << nrx) * sizeof(fr_t)), *ey = (fr_t *)malloc(((size_t)1 << nry) * sizeof(fr_t));
    orc_eq_evals(rx, nrx, ex); orc_eq_evals(ry, nry, ey);
    const orc_sparse *m[3] = {&I->A, &I->B, &I->C};
    for (int k = 0; k < 3; k++) {
        fr_t acc = FR_ZERO;
        for (size_t i = 0; i < m[k]->n; i++) {
            fr_t t; fr_mul(&t, &ex[m[k]->M[i].row], &ey[m[k]->M[i].col]); fr_mul(&t, &t, &m[k]->M[i].val); fr_add(&acc, &acc, &t);
        }
        ev[k] = acc;
    }
    free(ex); free(ey);
}

/* build z = vars || 1 || inputs || 0.. (r1csproof.rs prove) */
static fr_t *build_z(const orc_instance *I, const fr_t *vars, const fr_t *inputs, size_t ni) {
    size_t V = I->num_vars;
    fr_t *z = (fr_t *)calloc(2 * V, sizeof(fr_t));
    memcpy(z, vars, V * sizeof(fr_t)); z[V] = FR_ONE; memcpy(z + V + 1, inputs, ni * sizeof(fr_t));
    return z;
}

int orc_instance_is_sat(const orc_instance *I, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ni, int *sat) {
    if (nvars > I->num_vars) return ORC_ERR_INVALID_NUM_VARS;
    if (ni != I->num_inputs) return ORC_ERR_INVALID_NUM_INPUTS;
    fr_t *vars = (fr_t *)calloc(I->num_vars, sizeof(fr_t)), *inp = (fr_t *)calloc(ni + 1, sizeof(fr_t));
    for (size_t i = 0; i < nvars; i++) if (!fr_from_bytes(&vars[i], vars32 + 32 * i)) { free(vars); free(inp); return ORC_ERR_INVALID_SCALAR; }
    for (size_t i = 0; i < ni; i++) if (!fr_from_bytes(&inp[i], inputs32 + 32 * i)) { free(vars); free(inp); return ORC_ERR_INVALID_SCALAR; }
    fr_t *z = build_z(I, vars, inp, ni);
    size_t N = I->num_cons;
    fr_t *Az = (fr_t *)malloc(3 * N * sizeof(fr_t)), *Bz = Az + N, *Cz = Bz + N;
    orc_multiply_vec(I, z, Az, Bz, Cz);
    int ok = 1;
    for (size_t i = 0; i < N && ok; i++) { fr_t t; fr_mul(&t, &Az[i], &Bz[i]); if (!fr_eq(&t, &Cz[i])) ok = 0; }
    *sat = ok; free(Az); free(z); free(vars); free(inp);
    return ORC_OK;
}

/* ------------------------------------------------------------------ unipoly.rs */
void unipoly_from_evals(fr_t *c, const fr_t *e, size_t n) {
    fr_t two, six, two_inv, six_inv, t;
    fr_from_u64(&two, 2); fr_from_u64(&six, 6); fr_inv(&two_inv, &two); fr_inv(&six_inv, &six);
    if (n == 3) {
        /* c0 = e0 ; a = (e2 - 2 e1 + e0)/2 ; b = e1 - c0 - a  -> [c0, b, a] */
        fr_t a;
        fr_sub(&t, &e[2], &e[1]); fr_sub(&t, &t, &e[1]); fr_add(&t, &t, &e[0]); fr_mul(&a, &two_inv, &t);
        c[0] = e[0]; fr_sub(&t, &e[1], &e[0]); fr_sub(&c[1], &t, &a); c[2] = a;
    } else {
        /* d = e0 ; a = (e3 - 3 e2 + 3 e1 - e0)/6 ; b = (2 e0 - 5 e1 + 4 e2 - e3)/2 ; c = e1 - d - a - b -> [d, c, b, a] */
        fr_t a, b;
        fr_sub(&t, &e[3], &e[2]); fr_sub(&t, &t, &e[2]); fr_sub(&t, &t, &e[2]);
        fr_add(&t, &t, &e[1]); fr_add(&t, &t, &e[1]); fr_add(&t, &t, &e[1]); fr_sub(&t, &t, &e[0]); fr_mul(&a, &six_inv, &t);
        fr_add(&t, &e[0], &e[0]); for (int k = 0; k < 5; k++) fr_sub(&t, &t, &e[1]);
        for (int k = 0; k < 4; k++) fr_add(&t, &t, &e[2]);
        fr_sub(&t, &t, &e[3]); fr_mul(&b, &two_inv, &t);
        c[0] = e[0]; fr_sub(&t, &e[1], &e[0]); fr_sub(&t, &t, &a); fr_sub(&c[1], &t, &b); c[2] = b; c[3] = a;
    }
}
void unipoly_eval(fr_t *o, const fr_t *c, size_t n, const fr_t *r) {
    fr_t ev = c[0], pw = *r, t;
    for (size_t i = 1; i < n; i++) { fr_mul(&t, &pw, &c[i]); fr_add(&ev, &ev, &t); fr_mul(&pw, &pw, r); }
    *o = ev;
}

/* ------------------------------------------------------------------ random.rs RandomTape */
void tape_init(transcript_t *tape, const uint8_t seed32[32]) {
    /* upstream: Transcript::new(b"proof") + append_scalar(b"init_randomness", Scalar::random(OsRng)).
       Here the scalar is from_bytes_wide(seed || 0^32) so that proofs are reproducible. */
    uint8_t w[64]; memset(w, 0, 64); memcpy(w, seed32, 32);
    fr_t s; fr_from_bytes_wide(&s, w);
    tr_init(tape, "proof", 5);
    tr_append_scalar(tape, "init_randomness", &s);
}

void nizk_free(nizk_t *p) {
    free(p->comm_vars); free(p->sc1.comm_polys); free(p->sc1.comm_evals); free(p->sc1.proofs);
    free(p->sc2.comm_polys); free(p->sc2.comm_evals); free(p->sc2.proofs); free(p->pe.Lv); free(p->pe.Rv); free(p->rx); free(p->ry);
}

/* ------------------------------------------------------------------ nizk/mod.rs sigma protocols */
static void know_prove(know_t *pf, uint8_t C[32], const orc_mcgens *g, transcript_t *tr, transcript_t *tape, const fr_t *x, const fr_t *r) {
    tr_protocol_name(tr, "knowledge proof");
    fr_t t1, t2, c, t;
    tr_challenge_scalar(tape, "t1", &t1); tr_challenge_scalar(tape, "t2", &t2);
    commit_scalar_c(C, x, r, g); tr_append_point(tr, "C", C);
    commit_scalar_c(pf->alpha, &t1, &t2, g); tr_append_point(tr, "alpha", pf->alpha);
    tr_challenge_scalar(tr, "c", &c);
    fr_mul(&t, x, &c); fr_add(&pf->z1, &t, &t1);
    fr_mul(&t, r, &c); fr_add(&pf->z2, &t, &t2);
}
static int know_verify(const know_t *pf, const orc_mcgens *g, transcript_t *tr, const uint8_t C[32]) {
    tr_protocol_name(tr, "knowledge proof");
    tr_append_point(tr, "C", C); tr_append_point(tr, "alpha", pf->alpha);
    fr_t c; tr_challenge_scalar(tr, "c", &c);
    ge_t lhs, Cp, Ap, rhs;
    commit_scalar(&lhs, &pf->z1, &pf->z2, g);
    if (!ge_decode(&Cp, C) || !ge_decode(&Ap, pf->alpha)) return ORC_ERR_VERIFY_DECOMPRESS;
    ge_scalarmul(&rhs, &Cp, &c); ge_add(&rhs, &rhs, &Ap);
    return ge_eq(&lhs, &rhs) ? ORC_OK : ORC_ERR_VERIFY_INTERNAL;
}

static void eq_prove(eqp_t *pf, const orc_mcgens *g, transcript_t *tr, transcript_t *tape,
                     const fr_t *v1, const fr_t *s1, const fr_t *v2, const fr_t *s2) {
    tr_protocol_name(tr, "equality proof");
    fr_t r, c, t; uint8_t C1[32], C2[32];
    tr_challenge_scalar(tape, "r", &r);
    commit_scalar_c(C1, v1, s1, g); tr_append_point(tr, "C1", C1);
    commit_scalar_c(C2, v2, s2, g); tr_append_point(tr, "C2", C2);
    ge_t a; ge_scalarmul(&a, &g->h, &r); ge_encode(pf->alpha, &a); tr_append_point(tr, "alpha", pf->alpha);
    tr_challenge_scalar(tr, "c", &c);
    fr_sub(&t, s1, s2); fr_mul(&t, &c, &t); fr_add(&pf->z, &t, &r);
}
static int eq_verify(const eqp_t *pf, const orc_mcgens *g, transcript_t *tr, const uint8_t C1[32], const uint8_t C2[32]) {
    tr_protocol_name(tr, "equality proof");
    tr_append_point(tr, "C1", C1); tr_append_point(tr, "C2", C2); tr_append_point(tr, "alpha", pf->alpha);
    fr_t c; tr_challenge_scalar(tr, "c", &c);
    ge_t P1, P2, A, rhs, lhs;
    if (!ge_decode(&P1, C1) || !ge_decode(&P2, C2) || !ge_decode(&A, pf->alpha)) return ORC_ERR_VERIFY_DECOMPRESS;
    ge_sub(&rhs, &P1, &P2); ge_scalarmul(&rhs, &rhs, &c); ge_add(&rhs, &rhs, &A);
    ge_scalarmul(&lhs, &g->h, &pf->z);
    return ge_eq(&lhs, &rhs) ? ORC_OK : ORC_ERR_VERIFY_INTERNAL;
}

static void prod_prove(prod_t *pf, uint8_t X[32], uint8_t Y[32], uint8_t Z[32], const orc_mcgens *g, transcript_t *tr, transcript_t *tape,
                       const fr_t *x, const fr_t *rX, const fr_t *y, const fr_t *rY, const fr_t *z, const fr_t *rZ) {
    tr_protocol_name(tr, "product proof");
    fr_t b[5], c, t, u;
    tr_challenge_scalar(tape, "b1", &b[0]); tr_challenge_scalar(tape, "b2", &b[1]); tr_challenge_scalar(tape, "b3", &b[2]);
    tr_challenge_scalar(tape, "b4", &b[3]); tr_challenge_scalar(tape, "b5", &b[4]);
    ge_t Xp;
    commit_scalar(&Xp, x, rX, g); ge_encode(X, &Xp); tr_append_point(tr, "X", X);
    commit_scalar_c(Y, y, rY, g); tr_append_point(tr, "Y", Y);
    commit_scalar_c(Z, z, rZ, g); tr_append_point(tr, "Z", Z);
    commit_scalar_c(pf->alpha, &b[0], &b[1], g); tr_append_point(tr, "alpha", pf->alpha);
    commit_scalar_c(pf->beta, &b[2], &b[3], g); tr_append_point(tr, "beta", pf->beta);
    { orc_mcgens gx = {1, &Xp, g->h}; commit_scalar_c(pf->delta, &b[2], &b[4], &gx); } tr_append_point(tr, "delta", pf->delta);
    tr_challenge_scalar(tr, "c", &c);
    fr_mul(&t, &c, x); fr_add(&pf->z[0], &b[0], &t);
    fr_mul(&t, &c, rX); fr_add(&pf->z[1], &b[1], &t);
    fr_mul(&t, &c, y); fr_add(&pf->z[2], &b[2], &t);
    fr_mul(&t, &c, rY); fr_add(&pf->z[3], &b[3], &t);
    fr_mul(&u, rX, y); fr_sub(&u, rZ, &u); fr_mul(&t, &c, &u); fr_add(&pf->z[4], &b[4], &t);
}
static int prod_check(const uint8_t P[32], const uint8_t X[32], const fr_t *c, const orc_mcgens *g, const fr_t *z1, const fr_t *z2) {
    ge_t Pp, Xp, lhs, rhs;
    if (!ge_decode(&Pp, P) || !ge_decode(&Xp, X)) return 0;
    ge_scalarmul(&lhs, &Xp, c); ge_add(&lhs, &lhs, &Pp);
    commit_scalar(&rhs, z1, z2, g);
    return ge_eq(&lhs, &rhs);
}
static int prod_verify(const prod_t *pf, const orc_mcgens *g, transcript_t *tr, const uint8_t X[32], const uint8_t Y[32], const uint8_t Z[32]) {
    tr_protocol_name(tr, "product proof");
    tr_append_point(tr, "X", X); tr_append_point(tr, "Y", Y); tr_append_point(tr, "Z", Z);
    tr_append_point(tr, "alpha", pf->alpha); tr_append_point(tr, "beta", pf->beta); tr_append_point(tr, "delta", pf->delta);
    fr_t c; tr_challenge_scalar(tr, "c", &c);
    ge_t Xp; if (!ge_decode(&Xp, X)) return ORC_ERR_VERIFY_DECOMPRESS;
    orc_mcgens gx = {1, &Xp, g->h};
    int ok = prod_check(pf->alpha, X, &c, g, &pf->z[0], &pf->z[1]) && prod_check(pf->beta, Y, &c, g, &pf->z[2], &pf->z[3]) &&
             prod_check(pf->delta, Z, &c, &gx, &pf->z[2], &pf->z[4]);
    return ok ? ORC_OK : ORC_ERR_VERIFY_INTERNAL;
}

/* DotProductProof::prove (x_vec length n <= 4) */
static void dp_prove(dp_proof_t *pf, const orc_mcgens *g1, const orc_mcgens *gn, transcript_t *tr, transcript_t *tape,
                     const fr_t *x, size_t n, const fr_t *blind_x, const fr_t *a, const fr_t *y, const fr_t *blind_y) {
    tr_protocol_name(tr, "dot product proof");
    fr_t d[4], r_delta, r_beta, c, t, ad; uint8_t Cx[32], Cy[32];
    tr_challenge_vector(tape, "d_vec", d, n);
    tr_challenge_scalar(tape, "r_delta", &r_delta); tr_challenge_scalar(tape, "r_beta", &r_beta);
    commit_vec_c(Cx, x, n, blind_x, gn); tr_append_point(tr, "Cx", Cx);
    commit_scalar_c(Cy, y, blind_y, g1); tr_append_point(tr, "Cy", Cy);
    tr_append_scalars(tr, "a", a, n);
    commit_vec_c(pf->delta, d, n, &r_delta, gn); tr_append_point(tr, "delta", pf->delta);
    dot(&ad, a, d, n);
    commit_scalar_c(pf->beta, &ad, &r_beta, g1); tr_append_point(tr, "beta", pf->beta);
    tr_challenge_scalar(tr, "c", &c);
    pf->nz = n;
    for (size_t i = 0; i < n; i++) { fr_mul(&t, &c, &x[i]); fr_add(&pf->z[i], &t, &d[i]); }
    fr_mul(&t, &c, blind_x); fr_add(&pf->z_delta, &t, &r_delta);
    fr_mul(&t, &c, blind_y); fr_add(&pf->z_beta, &t, &r_beta);
}
static int dp_verify(const dp_proof_t *pf, const orc_mcgens *g1, const orc_mcgens *gn, transcript_t *tr,
                     const fr_t *a, size_t n, const uint8_t Cx[32], const uint8_t Cy[32]) {
    if (pf->nz != n || gn->n != n) return ORC_ERR_VERIFY_INTERNAL;
    tr_protocol_name(tr, "dot product proof");
    tr_append_point(tr, "Cx", Cx); tr_append_point(tr, "Cy", Cy);
    tr_append_scalars(tr, "a", a, n);
    tr_append_point(tr, "delta", pf->delta); tr_append_point(tr, "beta", pf->beta);
    fr_t c, za; tr_challenge_scalar(tr, "c", &c);
    ge_t X, Y, D, Bt, lhs, rhs;
    if (!ge_decode(&X, Cx) || !ge_decode(&Y, Cy) || !ge_decode(&D, pf->delta) || !ge_decode(&Bt, pf->beta)) return ORC_ERR_VERIFY_DECOMPRESS;
    ge_scalarmul(&lhs, &X, &c); ge_add(&lhs, &lhs, &D); commit_vec(&rhs, pf->z, n, &pf->z_delta, gn);
    int ok = ge_eq(&lhs, &rhs);
    dot(&za, pf->z, a, n);
    ge_scalarmul(&lhs, &Y, &c); ge_add(&lhs, &lhs, &Bt); commit_scalar(&rhs, &za, &pf->z_beta, g1);
    ok &= ge_eq(&lhs, &rhs);
    return ok ? ORC_OK : ORC_ERR_VERIFY_INTERNAL;
}

/* ------------------------------------------------------------------ sumcheck.rs ZKSumcheckInstanceProof */
void orc_sc_cubic_evals(const fr_t *A, const fr_t *B, const fr_t *C, const fr_t *D, size_t len2, fr_t e[3]) {
    size_t len = len2 / 2;
    fr_t e0 = FR_ZERO, e2 = FR_ZERO, e3 = FR_ZERO;
#pragma omp parallel num_threads(g_threads) if (len >= 2048)
    {
        fr_t l0 = FR_ZERO, l2 = FR_ZERO, l3 = FR_ZERO;
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < len; i++) {
            fr_t t, a, b, c, d, da, db, dc, dd;
            /* eval 0 */
            fr_mul(&t, &B[i], &C[i]); fr_sub(&t, &t, &D[i]); fr_mul(&t, &A[i], &t); fr_add(&l0, &l0, &t);
            /* eval 2: X(2) = 2 X[hi] - X[lo] */
            fr_sub(&da, &A[len + i], &A[i]); fr_sub(&db, &B[len + i], &B[i]); fr_sub(&dc, &C[len + i], &C[i]); fr_sub(&dd, &D[len + i], &D[i]);
            fr_add(&a, &A[len + i], &da); fr_add(&b, &B[len + i], &db); fr_add(&c, &C[len + i], &dc); fr_add(&d, &D[len + i], &dd);
            fr_mul(&t, &b, &c); fr_sub(&t, &t, &d); fr_mul(&t, &a, &t); fr_add(&l2, &l2, &t);
            /* eval 3: X(3) = X(2) + X[hi] - X[lo] */
            fr_add(&a, &a, &da); fr_add(&b, &b, &db); fr_add(&c, &c, &dc); fr_add(&d, &d, &dd);
            fr_mul(&t, &b, &c); fr_sub(&t, &t, &d); fr_mul(&t, &a, &t); fr_add(&l3, &l3, &t);
        }
#pragma omp critical
        { fr_add(&e0, &e0, &l0); fr_add(&e2, &e2, &l2); fr_add(&e3, &e3, &l3); }
    }
    e[0] = e0; e[1] = e2; e[2] = e3;
}
void orc_sc_quad_evals(const fr_t *A, const fr_t *B, size_t len2, fr_t e[2]) {
    size_t len = len2 / 2;
    fr_t e0 = FR_ZERO, e2 = FR_ZERO;
#pragma omp parallel num_threads(g_threads) if (len >= 2048)
    {
        fr_t l0 = FR_ZERO, l2 = FR_ZERO;
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < len; i++) {
            fr_t t, a, b;
            fr_mul(&t, &A[i], &B[i]); fr_add(&l0, &l0, &t);
            fr_add(&a, &A[len + i], &A[len + i]); fr_sub(&a, &a, &A[i]);
            fr_add(&b, &B[len + i], &B[len + i]); fr_sub(&b, &b, &B[i]);
            fr_mul(&t, &a, &b); fr_add(&l2, &l2, &t);
        }
#pragma omp critical
        { fr_add(&e0, &e0, &l0); fr_add(&e2, &e2, &l2); }
    }
    e[0] = e0; e[1] = e2;
}

typedef struct {
    fr_t claim; uint8_t comm_claim[32];
    const fr_t *blind_claim, *blinds_poly, *blinds_evals;
} sc_state_t;

static void zksc_alloc(zksc_t *p, size_t rounds) {
    p->rounds = rounds; p->comm_polys = (uint8_t *)malloc(32 * rounds); p->comm_evals = (uint8_t *)malloc(32 * rounds);
    p->proofs = (dp_proof_t *)calloc(rounds, sizeof(dp_proof_t));
}

/* the part of a round that is common to prove_quad / prove_cubic_with_additive_term, given the round's evals at 0,1,2(,3) */
static void sc_round(zksc_t *pf, size_t j, const fr_t *evals, size_t ne, sc_state_t *st, const orc_mcgens *g1, const orc_mcgens *gn,
                     transcript_t *tr, transcript_t *tape, fr_t *r_out) {
    fr_t poly[4], r_j, eval, w[2], target, blind, a[4], t, u;
    unipoly_from_evals(poly, evals, ne);
    uint8_t *comm_poly = pf->comm_polys + 32 * j, *comm_eval = pf->comm_evals + 32 * j;
    commit_vec_c(comm_poly, poly, ne, &st->blinds_poly[j], gn);
    tr_append_point(tr, "comm_poly", comm_poly);
    tr_challenge_scalar(tr, "challenge_nextround", &r_j);
    unipoly_eval(&eval, poly, ne, &r_j);
    commit_scalar_c(comm_eval, &eval, &st->blinds_evals[j], g1);
    tr_append_point(tr, "comm_claim_per_round", st->comm_claim);
    tr_append_point(tr, "comm_eval", comm_eval);
    tr_challenge_vector(tr, "combine_two_claims_to_one", w, 2);
    fr_mul(&t, &w[0], &st->claim); fr_mul(&u, &w[1], &eval); fr_add(&target, &t, &u);
    const fr_t *blind_sc = j == 0 ? st->blind_claim : &st->blinds_evals[j - 1];
    fr_mul(&t, &w[0], blind_sc); fr_mul(&u, &w[1], &st->blinds_evals[j]); fr_add(&blind, &t, &u);
    /* a = w0*(2,1,1,..) + w1*(1,r,r^2,..) */
    fr_t pw = FR_ONE, two; fr_from_u64(&two, 2);
    for (size_t i = 0; i < ne; i++) {
        fr_t asc = i == 0 ? two : FR_ONE;
        fr_mul(&t, &w[0], &asc); fr_mul(&u, &w[1], &pw); fr_add(&a[i], &t, &u);
        fr_mul(&pw, &pw, &r_j);
    }
    dp_prove(&pf->proofs[j], g1, gn, tr, tape, poly, ne, &st->blinds_poly[j], a, &target, &blind);
    st->claim = eval; memcpy(st->comm_claim, comm_eval, 32);
    *r_out = r_j;
}

/* ZKSumcheckInstanceProof::verify */
static int zksc_verify(const zksc_t *pf, const uint8_t comm_claim[32], size_t num_rounds, size_t degree, const orc_mcgens *g1,
                       const orc_mcgens *gn, transcript_t *tr, uint8_t comm_final[32], fr_t *r) {
    if (gn->n != degree + 1 || pf->rounds != num_rounds) return ORC_ERR_VERIFY_INTERNAL;
    size_t ne = degree + 1;
    for (size_t i = 0; i < num_rounds; i++) {
        const uint8_t *comm_poly = pf->comm_polys + 32 * i, *comm_eval = pf->comm_evals + 32 * i;
        const uint8_t *ccpr = i == 0 ? comm_claim : pf->comm_evals + 32 * (i - 1);
        tr_append_point(tr, "comm_poly", comm_poly);
        fr_t r_i, w[2], a[4], t, u; tr_challenge_scalar(tr, "challenge_nextround", &r_i);
        tr_append_point(tr, "comm_claim_per_round", ccpr); tr_append_point(tr, "comm_eval", comm_eval);
        tr_challenge_vector(tr, "combine_two_claims_to_one", w, 2);
        ge_t P0, P1, T; uint8_t comm_target[32];
        if (!ge_decode(&P0, ccpr) || !ge_decode(&P1, comm_eval)) return ORC_ERR_VERIFY_DECOMPRESS;
        ge_scalarmul(&P0, &P0, &w[0]); ge_scalarmul(&P1, &P1, &w[1]); ge_add(&T, &P0, &P1); ge_encode(comm_target, &T);
        fr_t pw = FR_ONE, two; fr_from_u64(&two, 2);
        for (size_t k = 0; k < ne; k++) {
            fr_t asc = k == 0 ? two : FR_ONE;
            fr_mul(&t, &w[0], &asc); fr_mul(&u, &w[1], &pw); fr_add(&a[k], &t, &u); fr_mul(&pw, &pw, &r_i);
        }
        int rc = dp_verify(&pf->proofs[i], g1, gn, tr, a, ne, comm_poly, comm_target);
        if (rc) return ORC_ERR_VERIFY_INTERNAL;
        r[i] = r_i;
    }
    memcpy(comm_final, pf->comm_evals + 32 * (num_rounds - 1), 32);
    return ORC_OK;
}

/* ------------------------------------------------------------------ nizk/bullet.rs BulletReductionProof */
static void bullet_prove(dplog_t *pf, transcript_t *tr, const ge_t *Q, const ge_t *Gv, const ge_t *H, const fr_t *a_in, const fr_t *b_in,
                         size_t n, const fr_t *blind, const fr_t *blinds /* 2*lg_n pairs, (v1[i], v2[i]) */,
                         fr_t *a_hat, fr_t *b_hat, ge_t *g_hat, fr_t *blind_fin_out) {
    size_t lg = ilog2(n);
    ge_t *G = (ge_t *)malloc(n * sizeof(ge_t)); memcpy(G, Gv, n * sizeof(ge_t));
    fr_t *a = (fr_t *)malloc(n * sizeof(fr_t)), *b = (fr_t *)malloc(n * sizeof(fr_t));
    memcpy(a, a_in, n * sizeof(fr_t)); memcpy(b, b_in, n * sizeof(fr_t));
    pf->n = lg; pf->Lv = (uint8_t *)malloc(32 * (lg ? lg : 1)); pf->Rv = (uint8_t *)malloc(32 * (lg ? lg : 1));
    fr_t blind_fin = *blind;
    fr_t *s = (fr_t *)malloc((n / 2 + 3) * sizeof(fr_t)); ge_t *P = (ge_t *)malloc((n / 2 + 3) * sizeof(ge_t));
    size_t round = 0;
    while (n != 1) {
        n /= 2;
        fr_t cL, cR, u, ui, t, x;
        dot(&cL, a, b + n, n); dot(&cR, a + n, b, n);
        const fr_t *bl = &blinds[2 * round], *br = &blinds[2 * round + 1];
        ge_t Lp, Rp;
        memcpy(s, a, n * sizeof(fr_t)); s[n] = cL; s[n + 1] = *bl; memcpy(P, G + n, n * sizeof(ge_t)); P[n] = *Q; P[n + 1] = *H;
        ge_msm(&Lp, s, P, n + 2);
        memcpy(s, a + n, n * sizeof(fr_t)); s[n] = cR; s[n + 1] = *br; memcpy(P, G, n * sizeof(ge_t)); P[n] = *Q; P[n + 1] = *H;
        ge_msm(&Rp, s, P, n + 2);
        ge_encode(pf->Lv + 32 * round, &Lp); ge_encode(pf->Rv + 32 * round, &Rp);
        tr_append_point(tr, "L", pf->Lv + 32 * round); tr_append_point(tr, "R", pf->Rv + 32 * round);
        tr_challenge_scalar(tr, "u", &u); fr_inv(&ui, &u);
#pragma omp parallel for num_threads(g_threads) schedule(static) if (n >= 64)
        for (size_t i = 0; i < n; i++) {
            fr_t p, q; ge_t g0, g1;
            fr_mul(&p, &a[i], &u); fr_mul(&q, &ui, &a[n + i]); fr_add(&a[i], &p, &q);
            fr_mul(&p, &b[i], &ui); fr_mul(&q, &u, &b[n + i]); fr_add(&b[i], &p, &q);
            ge_scalarmul(&g0, &G[i], &ui); ge_scalarmul(&g1, &G[n + i], &u); ge_add(&G[i], &g0, &g1);
        }
        fr_mul(&t, bl, &u); fr_mul(&t, &t, &u); fr_mul(&x, br, &ui); fr_mul(&x, &x, &ui);
        fr_add(&blind_fin, &blind_fin, &t); fr_add(&blind_fin, &blind_fin, &x);
        round++;
    }
    *a_hat = a[0]; *b_hat = b[0]; *g_hat = G[0]; *blind_fin_out = blind_fin;
    free(G); free(a); free(b); free(s); free(P);
}

/* The same reduction with the round challenges supplied by the caller and stopped after `rounds` rounds: what the kernel-level
   tests compare one GPU bullet round against (L, R of every round; the folded a, b; the folded generators, compressed). */
void orc_bullet_reduce(const orc_gens *g, const fr_t *a_in, const fr_t *b_in, size_t n, const fr_t *blinds /* (bL, bR) per round */,
                       const fr_t *us, size_t rounds, uint8_t *LR /* 64 * rounds */, fr_t *a_out, fr_t *b_out, uint8_t *G_out32 /* n >> rounds each */) {
    const ge_t *Q = &g->pc_1.G[0], *H = &g->pc_n.h;
    ge_t *G = (ge_t *)malloc(n * sizeof(ge_t)); memcpy(G, g->pc_n.G, n * sizeof(ge_t));
    fr_t *a = (fr_t *)malloc(n * sizeof(fr_t)), *b = (fr_t *)malloc(n * sizeof(fr_t));
    memcpy(a, a_in, n * sizeof(fr_t)); memcpy(b, b_in, n * sizeof(fr_t));
    fr_t *s = (fr_t *)malloc((n / 2 + 3) * sizeof(fr_t)); ge_t *P = (ge_t *)malloc((n / 2 + 3) * sizeof(ge_t));
    for (size_t round = 0; round < rounds && n != 1; round++) {
        n /= 2;
        fr_t cL, cR, u = us[round], ui;
        dot(&cL, a, b + n, n); dot(&cR, a + n, b, n);
        ge_t Lp, Rp;
        memcpy(s, a, n * sizeof(fr_t)); s[n] = cL; s[n + 1] = blinds[2 * round]; memcpy(P, G + n, n * sizeof(ge_t)); P[n] = *Q; P[n + 1] = *H;
        ge_msm(&Lp, s, P, n + 2);
        memcpy(s, a + n, n * sizeof(fr_t)); s[n] = cR; s[n + 1] = blinds[2 * round + 1]; memcpy(P, G, n * sizeof(ge_t)); P[n] = *Q; P[n + 1] = *H;
        ge_msm(&Rp, s, P, n + 2);
        ge_encode(LR + 64 * round, &Lp); ge_encode(LR + 64 * round + 32, &Rp);
        fr_inv(&ui, &u);
        for (size_t i = 0; i < n; i++) {
            fr_t p, q; ge_t g0, g1;
            fr_mul(&p, &a[i], &u); fr_mul(&q, &ui, &a[n + i]); fr_add(&a[i], &p, &q);
            fr_mul(&p, &b[i], &ui); fr_mul(&q, &u, &b[n + i]); fr_add(&b[i], &p, &q);
            ge_scalarmul(&g0, &G[i], &ui); ge_scalarmul(&g1, &G[n + i], &u); ge_add(&G[i], &g0, &g1);
        }
    }
    memcpy(a_out, a, n * sizeof(fr_t)); memcpy(b_out, b, n * sizeof(fr_t));
    for (size_t i = 0; i < n; i++) ge_encode(G_out32 + 32 * i, &G[i]);
    free(G); free(a); free(b); free(s); free(P);
}

/* BulletReductionProof::verify (with verification_scalars) */
static int bullet_verify(const dplog_t *pf, size_t n, const fr_t *a, transcript_t *tr, const ge_t *Gamma, const ge_t *G,
                         ge_t *g_hat, ge_t *Gamma_hat, fr_t *a_hat) {
    size_t lg = pf->n;
    if (lg >= 32 || n != ((size_t)1 << lg)) return ORC_ERR_VERIFY_INTERNAL;
    fr_t *ch = (fr_t *)malloc((lg + 1) * sizeof(fr_t)), *chi = (fr_t *)malloc((lg + 1) * sizeof(fr_t));
    for (size_t i = 0; i < lg; i++) { tr_append_point(tr, "L", pf->Lv + 32 * i); tr_append_point(tr, "R", pf->Rv + 32 * i); tr_challenge_scalar(tr, "u", &ch[i]); }
    fr_t allinv = FR_ONE;
    for (size_t i = 0; i < lg; i++) { fr_inv(&chi[i], &ch[i]); fr_mul(&allinv, &allinv, &chi[i]); }
    for (size_t i = 0; i < lg; i++) { fr_sqr(&ch[i], &ch[i]); fr_sqr(&chi[i], &chi[i]); }
    fr_t *s = (fr_t *)malloc(n * sizeof(fr_t));
    s[0] = allinv;
    for (size_t i = 1; i < n; i++) {
        size_t lg_i = 0; while (((size_t)2 << lg_i) <= i) lg_i++;
        size_t k = (size_t)1 << lg_i;
        fr_mul(&s[i], &s[i - k], &ch[(lg - 1) - lg_i]);
    }
    ge_msm(g_hat, s, G, n);
    dot(a_hat, a, s, n);
    int rc = ORC_OK;
    fr_t *sc = (fr_t *)malloc((2 * lg + 1) * sizeof(fr_t)); ge_t *P = (ge_t *)malloc((2 * lg + 1) * sizeof(ge_t));
    for (size_t i = 0; i < lg; i++) {
        if (!ge_decode(&P[i], pf->Lv + 32 * i) || !ge_decode(&P[lg + i], pf->Rv + 32 * i)) { rc = ORC_ERR_VERIFY_DECOMPRESS; break; }
        sc[i] = ch[i]; sc[lg + i] = chi[i];
    }
    if (!rc) { sc[2 * lg] = FR_ONE; P[2 * lg] = *Gamma; ge_msm(Gamma_hat, sc, P, 2 * lg + 1); }
    free(ch); free(chi); free(s); free(sc); free(P);
    return rc;
}

/* ------------------------------------------------------------------ nizk/mod.rs DotProductProofLog, dense_mlpoly.rs PolyEvalProof */
void dplog_prove(dplog_t *pf, uint8_t Cy_out[32], const orc_mcgens *gn, const orc_mcgens *g1, transcript_t *tr, transcript_t *tape,
                 const fr_t *x, const fr_t *blind_x, const fr_t *a, size_t n, const fr_t *y, const fr_t *blind_y) {
    tr_protocol_name(tr, "dot product proof (log)");
    size_t lg = ilog2(n);
    fr_t d, r_delta, r_beta, c, t, u;
    tr_challenge_scalar(tape, "d", &d);
    tr_challenge_scalar(tape, "r_delta", &r_delta);
    tr_challenge_scalar(tape, "r_delta", &r_beta);                       /* sic: upstream reuses the label */
    fr_t *v1 = (fr_t *)malloc((2 * lg + 1) * sizeof(fr_t)), *v2 = (fr_t *)malloc((2 * lg + 1) * sizeof(fr_t));
    tr_challenge_vector(tape, "blinds_vec_1", v1, 2 * lg);
    tr_challenge_vector(tape, "blinds_vec_2", v2, 2 * lg);
    fr_t *blinds = (fr_t *)malloc((4 * lg + 2) * sizeof(fr_t));
    for (size_t i = 0; i < 2 * lg; i++) { blinds[2 * i] = v1[i]; blinds[2 * i + 1] = v2[i]; }
    uint8_t Cx[32];
    commit_vec_c(Cx, x, n, blind_x, gn); tr_append_point(tr, "Cx", Cx);
    commit_scalar_c(Cy_out, y, blind_y, g1); tr_append_point(tr, "Cy", Cy_out);
    fr_t blind_Gamma; fr_add(&blind_Gamma, blind_x, blind_y);
    fr_t x_hat, a_hat, rhat; ge_t g_hat;
    bullet_prove(pf, tr, &g1->G[0], gn->G, &gn->h, x, a, n, &blind_Gamma, blinds, &x_hat, &a_hat, &g_hat, &rhat);
    fr_t y_hat; fr_mul(&y_hat, &x_hat, &a_hat);
    { orc_mcgens gh = {1, &g_hat, g1->h}; commit_scalar_c(pf->delta, &d, &r_delta, &gh); } tr_append_point(tr, "delta", pf->delta);
    commit_scalar_c(pf->beta, &d, &r_beta, g1); tr_append_point(tr, "beta", pf->beta);
    tr_challenge_scalar(tr, "c", &c);
    fr_mul(&t, &c, &y_hat); fr_add(&pf->z1, &d, &t);
    fr_mul(&t, &c, &rhat); fr_add(&t, &t, &r_beta); fr_mul(&u, &a_hat, &t); fr_add(&pf->z2, &u, &r_delta);
    free(v1); free(v2); free(blinds);
}
int dplog_verify(const dplog_t *pf, size_t n, const orc_mcgens *gn, const orc_mcgens *g1, transcript_t *tr, const fr_t *a, const uint8_t Cx[32], const uint8_t Cy[32]) {
    if (gn->n != n) return ORC_ERR_VERIFY_INTERNAL;
    tr_protocol_name(tr, "dot product proof (log)");
    tr_append_point(tr, "Cx", Cx); tr_append_point(tr, "Cy", Cy);
    ge_t X, Y, Gamma, g_hat, Gamma_hat, D, Bt; fr_t a_hat, c;
    if (!ge_decode(&X, Cx) || !ge_decode(&Y, Cy)) return ORC_ERR_VERIFY_DECOMPRESS;
    ge_add(&Gamma, &X, &Y);
    int rc = bullet_verify(pf, n, a, tr, &Gamma, gn->G, &g_hat, &Gamma_hat, &a_hat);
    if (rc) return rc;
    tr_append_point(tr, "delta", pf->delta); tr_append_point(tr, "beta", pf->beta);
    tr_challenge_scalar(tr, "c", &c);
    if (!ge_decode(&D, pf->delta) || !ge_decode(&Bt, pf->beta)) return ORC_ERR_VERIFY_DECOMPRESS;
    ge_t lhs, rhs, t;
    ge_scalarmul(&lhs, &Gamma_hat, &c); ge_add(&lhs, &lhs, &Bt); ge_scalarmul(&lhs, &lhs, &a_hat); ge_add(&lhs, &lhs, &D);
    ge_scalarmul(&t, &g1->G[0], &a_hat); ge_add(&t, &t, &g_hat); ge_scalarmul(&rhs, &t, &pf->z1);
    ge_scalarmul(&t, &g1->h, &pf->z2); ge_add(&rhs, &rhs, &t);
    return ge_eq(&lhs, &rhs) ? ORC_OK : ORC_ERR_VERIFY_INTERNAL;
}

/* ------------------------------------------------------------------ bincode layout of NIZK [RECALL; SURVEY App. A item 12] */
void wb(wbuf_t *w, const void *d, size_t n) {
    if (w->len + n > w->cap) { w->cap = (w->len + n) * 2 + 64; w->p = (uint8_t *)realloc(w->p, w->cap); }
    memcpy(w->p + w->len, d, n); w->len += n;
}
void wb_u64(wbuf_t *w, uint64_t x) { uint8_t b[8]; for (int i = 0; i < 8; i++) { b[i] = (uint8_t)x; x >>= 8; } wb(w, b, 8); }
/* upstream Scalar derives Serialize over its [u64;4] Montgomery limbs, so bincode carries Montgomery form */
void wb_fr(wbuf_t *w, const fr_t *x) { uint8_t b[32]; fr_mont_bytes(b, x); wb(w, b, 32); }
static void wb_zksc(wbuf_t *w, const zksc_t *s) {
    wb_u64(w, s->rounds); wb(w, s->comm_polys, 32 * s->rounds);
    wb_u64(w, s->rounds); wb(w, s->comm_evals, 32 * s->rounds);
    wb_u64(w, s->rounds);
    for (size_t i = 0; i < s->rounds; i++) {
        const dp_proof_t *d = &s->proofs[i];
        wb(w, d->delta, 32); wb(w, d->beta, 32); wb_u64(w, d->nz); for (size_t k = 0; k < d->nz; k++) wb_fr(w, &d->z[k]);
        wb_fr(w, &d->z_delta); wb_fr(w, &d->z_beta);
    }
}
void r1cs_serialize_body(wbuf_t *wp, const nizk_t *p);
static void nizk_serialize(const nizk_t *p, uint8_t **out, size_t *len) {
    wbuf_t w = {0, 0, 0};
    r1cs_serialize_body(&w, p);
    wb_u64(&w, p->nrx); for (size_t i = 0; i < p->nrx; i++) wb_fr(&w, &p->rx[i]);
    wb_u64(&w, p->nry); for (size_t i = 0; i < p->nry; i++) wb_fr(&w, &p->ry[i]);
    *out = w.p; *len = w.len;
}
/* R1CSProof's fields in bincode order */
void r1cs_serialize_body(wbuf_t *wp, const nizk_t *p) {
    wbuf_t w = *wp;
    wb_u64(&w, p->nC); wb(&w, p->comm_vars, 32 * p->nC);
    wb_zksc(&w, &p->sc1);
    wb(&w, p->claims2, 128);
    wb(&w, p->pok.alpha, 32); wb_fr(&w, &p->pok.z1); wb_fr(&w, &p->pok.z2);
    wb(&w, p->prod.alpha, 32); wb(&w, p->prod.beta, 32); wb(&w, p->prod.delta, 32); for (int i = 0; i < 5; i++) wb_fr(&w, &p->prod.z[i]);
    wb(&w, p->eq1.alpha, 32); wb_fr(&w, &p->eq1.z);
    wb_zksc(&w, &p->sc2);
    wb(&w, p->comm_vars_at_ry, 32);
    wb_u64(&w, p->pe.n); wb(&w, p->pe.Lv, 32 * p->pe.n); wb_u64(&w, p->pe.n); wb(&w, p->pe.Rv, 32 * p->pe.n);
    wb(&w, p->pe.delta, 32); wb(&w, p->pe.beta, 32); wb_fr(&w, &p->pe.z1); wb_fr(&w, &p->pe.z2);
    wb(&w, p->eq2.alpha, 32); wb_fr(&w, &p->eq2.z);
    *wp = w;
}

void rb(rbuf_t *r, void *d, size_t n) { if (r->bad || r->pos + n > r->len) { r->bad = 1; memset(d, 0, n); return; } memcpy(d, r->p + r->pos, n); r->pos += n; }
uint64_t rb_u64(rbuf_t *r) { uint8_t b[8]; rb(r, b, 8); uint64_t x = 0; for (int i = 7; i >= 0; i--) x = (x << 8) | b[i]; return x; }
void rb_fr(rbuf_t *r, fr_t *x) { uint8_t b[32]; rb(r, b, 32); if (!r->bad && !fr_from_mont_bytes(x, b)) r->bad = 1; }
uint8_t *rb_vec32(rbuf_t *r, size_t *n, size_t max) {
    uint64_t k = rb_u64(r); if (r->bad || k > max) { r->bad = 1; k = 0; }
    uint8_t *v = (uint8_t *)malloc(32 * (k ? k : 1)); rb(r, v, 32 * k); *n = (size_t)k; return v;
}
static void rb_zksc(rbuf_t *r, zksc_t *s) {
    size_t n1, n2; s->comm_polys = rb_vec32(r, &n1, 64); s->comm_evals = rb_vec32(r, &n2, 64);
    uint64_t n3 = rb_u64(r); if (n1 != n2 || n3 != n1) r->bad = 1;
    s->rounds = n1; s->proofs = (dp_proof_t *)calloc(n1 ? n1 : 1, sizeof(dp_proof_t));
    for (size_t i = 0; i < n1 && !r->bad; i++) {
        dp_proof_t *d = &s->proofs[i];
        rb(r, d->delta, 32); rb(r, d->beta, 32); uint64_t nz = rb_u64(r); if (nz > 4) { r->bad = 1; break; }
        d->nz = nz; for (size_t k = 0; k < nz; k++) rb_fr(r, &d->z[k]);
        rb_fr(r, &d->z_delta); rb_fr(r, &d->z_beta);
    }
}
void r1cs_parse_body(rbuf_t *rp, nizk_t *p);
static int nizk_parse(nizk_t *p, const uint8_t *buf, size_t len) {
    rbuf_t r = {buf, len, 0, 0}; memset(p, 0, sizeof *p);
    r1cs_parse_body(&r, p);
    uint64_t n = rb_u64(&r); if (n > 64) r.bad = 1; p->nrx = r.bad ? 0 : n; p->rx = (fr_t *)calloc(p->nrx + 1, sizeof(fr_t)); for (size_t i = 0; i < p->nrx; i++) rb_fr(&r, &p->rx[i]);
    n = rb_u64(&r); if (n > 64) r.bad = 1; p->nry = r.bad ? 0 : n; p->ry = (fr_t *)calloc(p->nry + 1, sizeof(fr_t)); for (size_t i = 0; i < p->nry; i++) rb_fr(&r, &p->ry[i]);
    if (r.bad || r.pos != r.len) return ORC_ERR_MALFORMED_PROOF;
    return ORC_OK;
}
void r1cs_parse_body(rbuf_t *rp, nizk_t *p) {
    rbuf_t r = *rp;
    p->comm_vars = rb_vec32(&r, &p->nC, (size_t)1 << 32);
    rb_zksc(&r, &p->sc1);
    rb(&r, p->claims2, 128);
    rb(&r, p->pok.alpha, 32); rb_fr(&r, &p->pok.z1); rb_fr(&r, &p->pok.z2);
    rb(&r, p->prod.alpha, 32); rb(&r, p->prod.beta, 32); rb(&r, p->prod.delta, 32); for (int i = 0; i < 5; i++) rb_fr(&r, &p->prod.z[i]);
    rb(&r, p->eq1.alpha, 32); rb_fr(&r, &p->eq1.z);
    rb_zksc(&r, &p->sc2);
    rb(&r, p->comm_vars_at_ry, 32);
    size_t nl, nr; p->pe.Lv = rb_vec32(&r, &nl, 64); p->pe.Rv = rb_vec32(&r, &nr, 64); if (nl != nr) r.bad = 1; p->pe.n = nl;
    rb(&r, p->pe.delta, 32); rb(&r, p->pe.beta, 32); rb_fr(&r, &p->pe.z1); rb_fr(&r, &p->pe.z2);
    rb(&r, p->eq2.alpha, 32); rb_fr(&r, &p->eq2.z);
    *rp = r;
}

/* ------------------------------------------------------------------ r1csproof.rs R1CSProof::prove */
/* vars: V padded elements, inputs: ni elements; the transcript already carries the caller's protocol names; fills P (incl. rx, ry) */
void r1cs_prove(const orc_instance *I, const fr_t *vars, const fr_t *inputs, size_t ni, const orc_gens *g, transcript_t *tr, transcript_t *tape,
                nizk_t *P, double st[7]) {
    size_t N = I->num_cons, V = I->num_vars;
    double t0;
    tr_protocol_name(tr, "R1CS proof");

    /* polycommit: DensePolynomial::commit */
    t0 = now_ms();
    size_t ell = ilog2(V), Lsz = (size_t)1 << (ell / 2), Rsz = (size_t)1 << (ell - ell / 2);
    fr_t *blinds_vars = (fr_t *)malloc(Lsz * sizeof(fr_t));
    tr_challenge_vector(tape, "poly_blinds", blinds_vars, Lsz);
    P->nC = Lsz; P->comm_vars = (uint8_t *)malloc(32 * Lsz);
    orc_commit_rows(vars, Lsz, Rsz, blinds_vars, &g->pc_n, P->comm_vars);
    tr_append(tr, "poly_commitment", (const uint8_t *)"poly_commitment_begin", 21);
    for (size_t i = 0; i < Lsz; i++) tr_append_point(tr, "poly_commitment_share", P->comm_vars + 32 * i);
    tr_append(tr, "poly_commitment", (const uint8_t *)"poly_commitment_end", 19);
    st[0] = now_ms() - t0;

    /* phase one */
    fr_t *z = build_z(I, vars, inputs, ni);
    size_t nrx = ilog2(N), nry = ilog2(2 * V);
    fr_t *tau = (fr_t *)malloc((nrx + 1) * sizeof(fr_t));
    tr_challenge_vector(tr, "challenge_tau", tau, nrx);
    fr_t *Tq = (fr_t *)malloc(4 * N * sizeof(fr_t)), *Az = Tq + N, *Bz = Az + N, *Cz = Bz + N;
    t0 = now_ms();
    orc_eq_evals(tau, nrx, Tq);
    orc_multiply_vec(I, z, Az, Bz, Cz);
    st[1] = now_ms() - t0;

    t0 = now_ms();
    P->nrx = nrx; P->rx = (fr_t *)malloc((nrx + 1) * sizeof(fr_t));
    fr_t *bp = (fr_t *)malloc((nrx + nry + 2) * sizeof(fr_t)), *be = (fr_t *)malloc((nrx + nry + 2) * sizeof(fr_t));
    fr_t blind_claim_postsc1;
    {
        tr_challenge_vector(tape, "blinds_poly", bp, nrx); tr_challenge_vector(tape, "blinds_evals", be, nrx);
        zksc_alloc(&P->sc1, nrx);
        sc_state_t s; s.claim = FR_ZERO; s.blind_claim = &FR_ZERO; s.blinds_poly = bp; s.blinds_evals = be;
        commit_scalar_c(s.comm_claim, &FR_ZERO, &FR_ZERO, &g->sc_1);
        size_t len = N;
        for (size_t j = 0; j < nrx; j++) {
            fr_t e[3], ev[4];
            orc_sc_cubic_evals(Tq, Az, Bz, Cz, len, e);
            ev[0] = e[0]; fr_sub(&ev[1], &s.claim, &e[0]); ev[2] = e[1]; ev[3] = e[2];
            sc_round(&P->sc1, j, ev, 4, &s, &g->sc_1, &g->sc_4, tr, tape, &P->rx[j]);
            orc_fold_top(Tq, len, &P->rx[j]); orc_fold_top(Az, len, &P->rx[j]); orc_fold_top(Bz, len, &P->rx[j]); orc_fold_top(Cz, len, &P->rx[j]);
            len /= 2;
        }
        blind_claim_postsc1 = be[nrx - 1];
    }
    st[2] = now_ms() - t0;

    fr_t tau_claim = Tq[0], Az_claim = Az[0], Bz_claim = Bz[0], Cz_claim = Cz[0];
    fr_t Az_blind, Bz_blind, Cz_blind, prod_blind, prod, t, u;
    tr_challenge_scalar(tape, "Az_blind", &Az_blind); tr_challenge_scalar(tape, "Bz_blind", &Bz_blind);
    tr_challenge_scalar(tape, "Cz_blind", &Cz_blind); tr_challenge_scalar(tape, "prod_Az_Bz_blind", &prod_blind);
    know_prove(&P->pok, P->claims2[2], &g->sc_1, tr, tape, &Cz_claim, &Cz_blind);
    fr_mul(&prod, &Az_claim, &Bz_claim);
    prod_prove(&P->prod, P->claims2[0], P->claims2[1], P->claims2[3], &g->sc_1, tr, tape, &Az_claim, &Az_blind, &Bz_claim, &Bz_blind, &prod, &prod_blind);
    tr_append_point(tr, "comm_Az_claim", P->claims2[0]); tr_append_point(tr, "comm_Bz_claim", P->claims2[1]);
    tr_append_point(tr, "comm_Cz_claim", P->claims2[2]); tr_append_point(tr, "comm_prod_Az_Bz_claims", P->claims2[3]);
    {
        fr_t blind_expected, claim_post;
        fr_sub(&t, &prod_blind, &Cz_blind); fr_mul(&blind_expected, &tau_claim, &t);
        fr_sub(&t, &prod, &Cz_claim); fr_mul(&claim_post, &t, &tau_claim);
        eq_prove(&P->eq1, &g->sc_1, tr, tape, &claim_post, &blind_expected, &claim_post, &blind_claim_postsc1);
    }

    /* phase two */
    fr_t rA, rB, rC, claim2, blind_claim2;
    tr_challenge_scalar(tr, "challenege_Az", &rA); tr_challenge_scalar(tr, "challenege_Bz", &rB); tr_challenge_scalar(tr, "challenege_Cz", &rC);
    fr_mul(&claim2, &rA, &Az_claim); fr_mul(&t, &rB, &Bz_claim); fr_add(&claim2, &claim2, &t); fr_mul(&t, &rC, &Cz_claim); fr_add(&claim2, &claim2, &t);
    fr_mul(&blind_claim2, &rA, &Az_blind); fr_mul(&t, &rB, &Bz_blind); fr_add(&blind_claim2, &blind_claim2, &t); fr_mul(&t, &rC, &Cz_blind); fr_add(&blind_claim2, &blind_claim2, &t);
    t0 = now_ms();
    fr_t *ABC = (fr_t *)malloc(2 * V * sizeof(fr_t));
    {
        fr_t *erx = Tq;                                        /* reuse: N entries */
        orc_eq_evals(P->rx, nrx, erx);
        fr_t *eA = (fr_t *)malloc(3 * 2 * V * sizeof(fr_t)), *eB = eA + 2 * V, *eC = eB + 2 * V;
        orc_eval_table_sparse(I, erx, eA, eB, eC);
#pragma omp parallel for num_threads(g_threads) schedule(static) if (V >= 2048)
        for (size_t i = 0; i < 2 * V; i++) {
            fr_t a, b, c; fr_mul(&a, &rA, &eA[i]); fr_mul(&b, &rB, &eB[i]); fr_mul(&c, &rC, &eC[i]); fr_add(&a, &a, &b); fr_add(&ABC[i], &a, &c);
        }
        free(eA);
    }
    st[3] = now_ms() - t0;

    t0 = now_ms();
    P->nry = nry; P->ry = (fr_t *)malloc((nry + 1) * sizeof(fr_t));
    fr_t claims_phase2[2], blind_claim_postsc2;
    {
        tr_challenge_vector(tape, "blinds_poly", bp, nry); tr_challenge_vector(tape, "blinds_evals", be, nry);
        zksc_alloc(&P->sc2, nry);
        sc_state_t s; s.claim = claim2; s.blind_claim = &blind_claim2; s.blinds_poly = bp; s.blinds_evals = be;
        commit_scalar_c(s.comm_claim, &claim2, &blind_claim2, &g->sc_1);
        size_t len = 2 * V;
        for (size_t j = 0; j < nry; j++) {
            fr_t e[2], ev[3];
            orc_sc_quad_evals(z, ABC, len, e);
            ev[0] = e[0]; fr_sub(&ev[1], &s.claim, &e[0]); ev[2] = e[1];
            sc_round(&P->sc2, j, ev, 3, &s, &g->sc_1, &g->sc_3, tr, tape, &P->ry[j]);
            orc_fold_top(z, len, &P->ry[j]); orc_fold_top(ABC, len, &P->ry[j]);
            len /= 2;
        }
        claims_phase2[0] = z[0]; claims_phase2[1] = ABC[0];
        blind_claim_postsc2 = be[nry - 1];
    }
    st[4] = now_ms() - t0;

    /* polyeval: poly_vars.evaluate(ry[1..]) + PolyEvalProof::prove */
    t0 = now_ms();
    fr_t eval_vars_at_ry, blind_eval;
    {
        const fr_t *r = P->ry + 1; size_t rl = nry - 1;                 /* = ell */
        fr_t *chis = (fr_t *)malloc(V * sizeof(fr_t));
        orc_eq_evals(r, rl, chis); dot(&eval_vars_at_ry, vars, chis, V);
        free(chis);
        tr_challenge_scalar(tape, "blind_eval", &blind_eval);
        tr_protocol_name(tr, "polynomial evaluation proof");
        size_t lv = rl / 2;
        fr_t *Lv = (fr_t *)malloc(Lsz * sizeof(fr_t)), *Rv = (fr_t *)malloc(Rsz * sizeof(fr_t)), *LZ = (fr_t *)malloc(Rsz * sizeof(fr_t));
        orc_eq_evals(r, lv, Lv); orc_eq_evals(r + lv, rl - lv, Rv);
        orc_poly_bound(vars, Lsz, Rsz, Lv, LZ);
        fr_t LZ_blind; dot(&LZ_blind, blinds_vars, Lv, Lsz);
        dplog_prove(&P->pe, P->comm_vars_at_ry, &g->pc_n, &g->pc_1, tr, tape, LZ, &LZ_blind, Rv, Rsz, &eval_vars_at_ry, &blind_eval);
        free(Lv); free(Rv); free(LZ);
    }
    st[5] = now_ms() - t0;
    {
        fr_t one_m, blind_eval_Z, blind_expected, claim_post;
        fr_sub(&one_m, &FR_ONE, &P->ry[0]); fr_mul(&blind_eval_Z, &one_m, &blind_eval);
        fr_mul(&blind_expected, &claims_phase2[1], &blind_eval_Z);
        fr_mul(&claim_post, &claims_phase2[0], &claims_phase2[1]);
        eq_prove(&P->eq2, &g->pc_1, tr, tape, &claim_post, &blind_expected, &claim_post, &blind_claim_postsc2);
    }
    (void)u;
    free(blinds_vars); free(z); free(tau); free(Tq); free(bp); free(be); free(ABC);
}

/* ------------------------------------------------------------------ lib.rs NIZK::prove */
int orc_nizk_prove(const orc_instance *I, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ni, const orc_gens *g,
                   const uint8_t *tlabel, size_t tlabel_len, const uint8_t seed32[32], uint8_t **proof, size_t *proof_len, double *ms) {
    size_t V = I->num_vars;
    if (nvars > V) return ORC_ERR_INVALID_NUM_VARS;
    if (ni != I->num_inputs) return ORC_ERR_INVALID_NUM_INPUTS;
    fr_t *vars = (fr_t *)calloc(V, sizeof(fr_t)), *inputs = (fr_t *)calloc(ni + 1, sizeof(fr_t));     /* VarsAssignment::pad */
    for (size_t i = 0; i < nvars; i++) if (!fr_from_bytes(&vars[i], vars32 + 32 * i)) { free(vars); free(inputs); return ORC_ERR_INVALID_SCALAR; }
    for (size_t i = 0; i < ni; i++) if (!fr_from_bytes(&inputs[i], inputs32 + 32 * i)) { free(vars); free(inputs); return ORC_ERR_INVALID_SCALAR; }
    double t_start = now_ms(), st[7] = {0};
    transcript_t tr, tape; nizk_t P; memset(&P, 0, sizeof P);
    tr_init(&tr, (const char *)tlabel, tlabel_len);
    tape_init(&tape, seed32);
    tr_protocol_name(&tr, "Spartan NIZK proof");
    r1cs_prove(I, vars, inputs, ni, g, &tr, &tape, &P, st);
    nizk_serialize(&P, proof, proof_len);
    st[6] = now_ms() - t_start;
    if (ms) memcpy(ms, st, sizeof st);
    nizk_free(&P); free(vars); free(inputs);
    return ORC_OK;
}

/* ------------------------------------------------------------------ r1csproof.rs R1CSProof::verify */
/* returns the challenges (rx, ry) the transcript produced; inst_evals = (A, B, C)(rx, ry) as claimed by the caller */
int r1cs_verify(const nizk_t *P, size_t N, size_t V, const fr_t *inputs, size_t ni, const fr_t inst_evals[3], const orc_gens *g, transcript_t *tr,
                fr_t *rx, fr_t *ry) {
    size_t nrx = ilog2(N), nry = ilog2(2 * V);
    int rc = ORC_OK;
    fr_t *tau = (fr_t *)malloc((nrx + 1) * sizeof(fr_t));
    fr_t *Lv = NULL, *Rv = NULL; ge_t *Cs = NULL;
    size_t ell = ilog2(V), Lsz = (size_t)1 << (ell / 2), Rsz = (size_t)1 << (ell - ell / 2);
    if (P->nC != Lsz || P->pe.n != ilog2(Rsz)) { rc = ORC_ERR_VERIFY_INTERNAL; goto done; }
    tr_protocol_name(tr, "R1CS proof");
    tr_append(tr, "poly_commitment", (const uint8_t *)"poly_commitment_begin", 21);
    for (size_t i = 0; i < P->nC; i++) tr_append_point(tr, "poly_commitment_share", P->comm_vars + 32 * i);
    tr_append(tr, "poly_commitment", (const uint8_t *)"poly_commitment_end", 19);
    tr_challenge_vector(tr, "challenge_tau", tau, nrx);

    uint8_t claim_phase1[32], comm_post1[32], comm_post2[32];
    commit_scalar_c(claim_phase1, &FR_ZERO, &FR_ZERO, &g->sc_1);
    if ((rc = zksc_verify(&P->sc1, claim_phase1, nrx, 3, &g->sc_1, &g->sc_4, tr, comm_post1, rx))) goto done;
    const uint8_t *cAz = P->claims2[0], *cBz = P->claims2[1], *cCz = P->claims2[2], *cPr = P->claims2[3];
    if ((rc = know_verify(&P->pok, &g->sc_1, tr, cCz))) goto done;
    if ((rc = prod_verify(&P->prod, &g->sc_1, tr, cAz, cBz, cPr))) goto done;
    tr_append_point(tr, "comm_Az_claim", cAz); tr_append_point(tr, "comm_Bz_claim", cBz);
    tr_append_point(tr, "comm_Cz_claim", cCz); tr_append_point(tr, "comm_prod_Az_Bz_claims", cPr);
    fr_t taus_bound = FR_ONE, t, u, om;
    for (size_t i = 0; i < nrx; i++) {
        fr_mul(&t, &rx[i], &tau[i]); fr_sub(&om, &FR_ONE, &rx[i]); fr_sub(&u, &FR_ONE, &tau[i]); fr_mul(&u, &om, &u); fr_add(&t, &t, &u);
        fr_mul(&taus_bound, &taus_bound, &t);
    }
    ge_t Ppr, Pcz, Paz, Pbz, E; uint8_t expected1[32];
    if (!ge_decode(&Ppr, cPr) || !ge_decode(&Pcz, cCz) || !ge_decode(&Paz, cAz) || !ge_decode(&Pbz, cBz)) { rc = ORC_ERR_VERIFY_DECOMPRESS; goto done; }
    ge_sub(&E, &Ppr, &Pcz); ge_scalarmul(&E, &E, &taus_bound); ge_encode(expected1, &E);
    if ((rc = eq_verify(&P->eq1, &g->sc_1, tr, expected1, comm_post1))) goto done;

    fr_t rA, rB, rC;
    tr_challenge_scalar(tr, "challenege_Az", &rA); tr_challenge_scalar(tr, "challenege_Bz", &rB); tr_challenge_scalar(tr, "challenege_Cz", &rC);
    uint8_t comm_claim2[32];
    { ge_t a, b, c; ge_scalarmul(&a, &Paz, &rA); ge_scalarmul(&b, &Pbz, &rB); ge_scalarmul(&c, &Pcz, &rC); ge_add(&a, &a, &b); ge_add(&a, &a, &c); ge_encode(comm_claim2, &a); }
    if ((rc = zksc_verify(&P->sc2, comm_claim2, nry, 2, &g->sc_1, &g->sc_3, tr, comm_post2, ry))) goto done;

    /* PolyEvalProof::verify */
    {
        const fr_t *r = ry + 1; size_t rl = nry - 1, lv = rl / 2;
        tr_protocol_name(tr, "polynomial evaluation proof");
        Lv = (fr_t *)malloc(Lsz * sizeof(fr_t)); Rv = (fr_t *)malloc(Rsz * sizeof(fr_t));
        orc_eq_evals(r, lv, Lv); orc_eq_evals(r + lv, rl - lv, Rv);
        Cs = (ge_t *)malloc(Lsz * sizeof(ge_t));
        for (size_t i = 0; i < Lsz; i++) if (!ge_decode(&Cs[i], P->comm_vars + 32 * i)) { rc = ORC_ERR_VERIFY_DECOMPRESS; goto done; }
        ge_t CLZ; uint8_t C_LZ[32]; ge_msm(&CLZ, Lv, Cs, Lsz); ge_encode(C_LZ, &CLZ);
        if ((rc = dplog_verify(&P->pe, Rsz, &g->pc_n, &g->pc_1, tr, Rv, C_LZ, P->comm_vars_at_ry))) goto done;
    }
    /* poly_input_eval: SparsePolynomial over (1, inputs) at ry[1..], MSB-first bits */
    fr_t poly_input_eval = FR_ZERO;
    {
        size_t nb = ilog2(V);
        for (size_t idx = 0; idx <= ni; idx++) {
            fr_t chi = FR_ONE;
            for (size_t j = 0; j < nb; j++) {
                int bit = (idx >> (nb - j - 1)) & 1;
                if (bit) fr_mul(&chi, &chi, &ry[1 + j]); else { fr_sub(&om, &FR_ONE, &ry[1 + j]); fr_mul(&chi, &chi, &om); }
            }
            fr_t val = idx == 0 ? FR_ONE : inputs[idx - 1];
            fr_mul(&t, &chi, &val); fr_add(&poly_input_eval, &poly_input_eval, &t);
        }
    }
    {
        ge_t Cv, Ci, Z; uint8_t expected2[32];
        if (!ge_decode(&Cv, P->comm_vars_at_ry)) { rc = ORC_ERR_VERIFY_DECOMPRESS; goto done; }
        commit_scalar(&Ci, &poly_input_eval, &FR_ZERO, &g->pc_1);
        fr_sub(&om, &FR_ONE, &ry[0]);
        ge_scalarmul(&Cv, &Cv, &om); ge_scalarmul(&Ci, &Ci, &ry[0]); ge_add(&Z, &Cv, &Ci);
        fr_mul(&t, &rA, &inst_evals[0]); fr_mul(&u, &rB, &inst_evals[1]); fr_add(&t, &t, &u); fr_mul(&u, &rC, &inst_evals[2]); fr_add(&t, &t, &u);
        ge_scalarmul(&Z, &Z, &t); ge_encode(expected2, &Z);
        if ((rc = eq_verify(&P->eq2, &g->sc_1, tr, expected2, comm_post2))) goto done;
    }
done:
    free(tau); free(Lv); free(Rv); free(Cs);
    return rc;
}

/* ------------------------------------------------------------------ lib.rs NIZK::verify */
int orc_nizk_verify(const orc_instance *I, const uint8_t *inputs32, size_t ni, const orc_gens *g, const uint8_t *tlabel, size_t tlabel_len,
                    const uint8_t *proof, size_t proof_len) {
    size_t N = I->num_cons, V = I->num_vars, nrx = ilog2(N), nry = ilog2(2 * V);
    if (ni != I->num_inputs) return ORC_ERR_INVALID_NUM_INPUTS;
    nizk_t P; int rc = nizk_parse(&P, proof, proof_len);
    fr_t *inputs = (fr_t *)calloc(ni + 1, sizeof(fr_t));
    fr_t *rx = (fr_t *)malloc((nrx + 1) * sizeof(fr_t)), *ry = (fr_t *)malloc((nry + 1) * sizeof(fr_t));
    if (rc) goto done;
    for (size_t i = 0; i < ni; i++) if (!fr_from_bytes(&inputs[i], inputs32 + 32 * i)) { rc = ORC_ERR_INVALID_SCALAR; goto done; }
    if (P.nrx != nrx || P.nry != nry) { rc = ORC_ERR_VERIFY_INTERNAL; goto done; }
    transcript_t tr; tr_init(&tr, (const char *)tlabel, tlabel_len);
    tr_protocol_name(&tr, "Spartan NIZK proof");
    fr_t inst_evals[3];
    inst_evaluate(I, P.rx, nrx, P.ry, nry, inst_evals);
    if ((rc = r1cs_verify(&P, N, V, inputs, ni, inst_evals, g, &tr, rx, ry))) goto done;
    /* NIZK::verify: claimed (rx, ry) must be the ones the transcript produced */
    for (size_t i = 0; i < nrx; i++) if (!fr_eq(&rx[i], &P.rx[i])) { rc = ORC_ERR_VERIFY_INTERNAL; goto done; }
    for (size_t i = 0; i < nry; i++) if (!fr_eq(&ry[i], &P.ry[i])) { rc = ORC_ERR_VERIFY_INTERNAL; goto done; }
done:
    nizk_free(&P); free(inputs); free(rx); free(ry);
    return rc;
}
