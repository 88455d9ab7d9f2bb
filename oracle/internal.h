/*
 * ORACLE — test infrastructure only (see fr.h).  Internals shared by the two translation units of the CPU restatement:
 * spartan.c (R1CS satisfiability proof, NIZK) and snark.c (computation commitment, R1CSEvalProof, SNARK).
 */
#ifndef OTTI_ORACLE_INTERNAL_H
#define OTTI_ORACLE_INTERNAL_H
#include "spartan.h"

/* ------------------------------------------------------------------ proof structures (field order = bincode order) */
typedef struct { uint8_t delta[32], beta[32]; size_t nz; fr_t z[4]; fr_t z_delta, z_beta; } dp_proof_t;
typedef struct { size_t rounds; uint8_t *comm_polys, *comm_evals; dp_proof_t *proofs; } zksc_t;
typedef struct { uint8_t alpha[32]; fr_t z1, z2; } know_t;
typedef struct { uint8_t alpha[32], beta[32], delta[32]; fr_t z[5]; } prod_t;
typedef struct { uint8_t alpha[32]; fr_t z; } eqp_t;
typedef struct { size_t n; uint8_t *Lv, *Rv; uint8_t delta[32], beta[32]; fr_t z1, z2; } dplog_t;
typedef struct {
    size_t nC; uint8_t *comm_vars;
    zksc_t sc1;
    uint8_t claims2[4][32];
    know_t pok; prod_t prod;
    eqp_t eq1;
    zksc_t sc2;
    uint8_t comm_vars_at_ry[32];
    dplog_t pe;
    eqp_t eq2;
    size_t nrx, nry; fr_t *rx, *ry;
} nizk_t;

typedef struct { uint8_t *p; size_t len, cap; } wbuf_t;
typedef struct { const uint8_t *p; size_t len, pos; int bad; } rbuf_t;

extern int g_threads;
double now_ms(void);
size_t ilog2(size_t n);
size_t next_pow2(size_t n);
void gens_stream(ge_t *out, size_t count, const char *label);
void mcgens_from(orc_mcgens *g, const ge_t *P, size_t n, const ge_t *h);
void mcgens_free(orc_mcgens *g);
void commit_vec(ge_t *o, const fr_t *v, size_t n, const fr_t *blind, const orc_mcgens *g);
void commit_scalar(ge_t *o, const fr_t *x, const fr_t *blind, const orc_mcgens *g);
void commit_vec_c(uint8_t out[32], const fr_t *v, size_t n, const fr_t *blind, const orc_mcgens *g);
void commit_scalar_c(uint8_t out[32], const fr_t *x, const fr_t *blind, const orc_mcgens *g);
void dot(fr_t *o, const fr_t *a, const fr_t *b, size_t n);
void inst_evaluate(const orc_instance *I, const fr_t *rx, size_t nrx, const fr_t *ry, size_t nry, fr_t ev[3]);
void unipoly_from_evals(fr_t *c, const fr_t *e, size_t n);
void unipoly_eval(fr_t *o, const fr_t *c, size_t n, const fr_t *r);
void tape_init(transcript_t *tape, const uint8_t seed32[32]);
/* nizk/mod.rs DotProductProofLog over (gens_n, gens_1) */
void dplog_prove(dplog_t *pf, uint8_t Cy_out[32], const orc_mcgens *gn, const orc_mcgens *g1, transcript_t *tr, transcript_t *tape,
                 const fr_t *x, const fr_t *blind_x, const fr_t *a, size_t n, const fr_t *y, const fr_t *blind_y);
int dplog_verify(const dplog_t *pf, size_t n, const orc_mcgens *gn, const orc_mcgens *g1, transcript_t *tr, const fr_t *a, const uint8_t Cx[32], const uint8_t Cy[32]);
void wb(wbuf_t *w, const void *d, size_t n);
void wb_u64(wbuf_t *w, uint64_t x);
void wb_fr(wbuf_t *w, const fr_t *x);
void rb(rbuf_t *r, void *d, size_t n);
uint64_t rb_u64(rbuf_t *r);
void rb_fr(rbuf_t *r, fr_t *x);
uint8_t *rb_vec32(rbuf_t *r, size_t *n, size_t max);
/* r1csproof.rs R1CSProof */
void r1cs_prove(const orc_instance *I, const fr_t *vars, const fr_t *inputs, size_t ni, const orc_gens *g, transcript_t *tr, transcript_t *tape,
                nizk_t *P, double st[7]);
int r1cs_verify(const nizk_t *P, size_t N, size_t V, const fr_t *inputs, size_t ni, const fr_t inst_evals[3], const orc_gens *g, transcript_t *tr,
                fr_t *rx, fr_t *ry);
void r1cs_serialize_body(wbuf_t *w, const nizk_t *p);
void r1cs_parse_body(rbuf_t *r, nizk_t *p);
void nizk_free(nizk_t *p);
#endif
