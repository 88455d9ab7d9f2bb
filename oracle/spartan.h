/*
 * ORACLE — test infrastructure only (see fr.h).  CPU restatement of libspartan's NIZK (R1CS satisfiability) prover and
 * verifier, the path `spzk verify --nizk` reaches [REF /root/reference/run.py:58, run.py:100].
 *
 * PARITY UNPINNED: the reference's Spartan/ and spartan-zkinterface/ submodules are empty directories and no golden
 * proof bytes exist anywhere in /root/reference; this restates upstream microsoft/Spartan (fork elefthei/Spartan,
 * pinned commit unknown) from its published protocol — SURVEY.md App. A [RECALL].  Every function cites the upstream
 * file/fn it follows.  Primitives are pinned by SURVEY.md App. B known answers + libsodium fixtures in tests/golden/.
 */
#ifndef OTTI_ORACLE_SPARTAN_H
#define OTTI_ORACLE_SPARTAN_H
#include <stdint.h>
#include <stddef.h>
#include "fr.h"
#include "ristretto.h"
#include "merlin.h"

#ifdef __cplusplus
extern "C" {
#endif

/* same layout as include/otti_spartan.h otti_entry (plain data, no shared header on purpose) */
typedef struct { uint64_t row, col; uint8_t val[32]; } orc_entry;

/* error codes mirror upstream R1CSError / ProofVerifyError [RECALL src/errors.rs] */
enum {
    ORC_OK = 0,
    ORC_ERR_NON_POW2_CONS = -1, ORC_ERR_NON_POW2_VARS = -2, ORC_ERR_INVALID_NUM_INPUTS = -3,
    ORC_ERR_INVALID_NUM_VARS = -4, ORC_ERR_INVALID_SCALAR = -5, ORC_ERR_INVALID_INDEX = -6,
    ORC_ERR_VERIFY_INTERNAL = -10, ORC_ERR_VERIFY_DECOMPRESS = -11, ORC_ERR_MALFORMED_PROOF = -12
};

typedef struct { size_t n; ge_t *G; ge_t h; } orc_mcgens;                 /* commitments.rs MultiCommitGens */
typedef struct { size_t row, col; fr_t val; } orc_mentry;
typedef struct { size_t n; orc_mentry *M; } orc_sparse;
typedef struct {
    size_t num_cons, num_vars, num_inputs;                                /* padded cons/vars */
    orc_sparse A, B, C;
} orc_instance;
typedef struct {
    orc_mcgens pc_n, pc_1;                                                /* gens_pc.gens.{gens_n,gens_1} */
    orc_mcgens sc_1, sc_3, sc_4;                                          /* gens_sc */
} orc_gens;

void orc_set_threads(int n);
int  orc_get_threads(void);

int  orc_instance_new(size_t num_cons, size_t num_vars, size_t num_inputs,
                      const orc_entry *A, size_t nA, const orc_entry *B, size_t nB, const orc_entry *C, size_t nC,
                      orc_instance **out);
void orc_instance_free(orc_instance *);
int  orc_instance_is_sat(const orc_instance *, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs, int *sat);
orc_gens *orc_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs);
void orc_gens_free(orc_gens *);

/* stage_ms: polycommit, multiply_vec, sc_phase_one, eval_table_sparse, sc_phase_two, polyeval, total (7 doubles) or NULL */
int  orc_nizk_prove(const orc_instance *, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs,
                    const orc_gens *, const uint8_t *tlabel, size_t tlabel_len, const uint8_t seed32[32],
                    uint8_t **proof, size_t *proof_len, double *stage_ms);
int  orc_nizk_verify(const orc_instance *, const uint8_t *inputs32, size_t ninputs, const orc_gens *,
                     const uint8_t *tlabel, size_t tlabel_len, const uint8_t *proof, size_t proof_len);
void orc_buf_free(void *);

/* ---- SNARK mode (snark.c): lib.rs SNARKGens / SNARK::{encode, prove, verify}, r1csinstance.rs R1CSCommitment / R1CSEvalProof ---- */
typedef struct orc_snark_gens orc_snark_gens;
typedef struct orc_snark_comm orc_snark_comm;                             /* ComputationCommitment (+ ComputationDecommitment on the prover side) */
orc_snark_gens *orc_snark_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs, size_t num_nz_entries);
void orc_snark_gens_free(orc_snark_gens *);
const orc_gens *orc_snark_gens_sat(const orc_snark_gens *);
orc_snark_comm *orc_snark_encode(const orc_instance *, const orc_snark_gens *);      /* NULL if the generators were made for another size */
void orc_snark_comm_bytes(const orc_snark_comm *, uint8_t **out, size_t *len);       /* bincode of the commitment */
orc_snark_comm *orc_snark_comm_parse(const uint8_t *buf, size_t len);                /* the verifier's view */
void orc_snark_comm_free(orc_snark_comm *);
/* stage_ms: 9 doubles or NULL — the seven R1CSProof stages, [7] R1CSEvalProof, [8] total */
int  orc_snark_prove(const orc_instance *, const orc_snark_comm *, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs,
                     const orc_snark_gens *, const uint8_t *tlabel, size_t tlabel_len, const uint8_t seed32[32], uint8_t **proof, size_t *proof_len, double *stage_ms);
int  orc_snark_verify(const orc_snark_comm *, const uint8_t *inputs32, size_t ninputs, const orc_snark_gens *, const uint8_t *tlabel, size_t tlabel_len,
                      const uint8_t *proof, size_t proof_len);

/* ---- kernel-level restatements (fr_t arrays are Montgomery form, as stored by fr.h) ---- */
void orc_eq_evals(const fr_t *r, size_t ell, fr_t *out /* 2^ell */);                           /* dense_mlpoly.rs EqPolynomial::evals */
void orc_multiply_vec(const orc_instance *, const fr_t *z /* 2V */, fr_t *Az, fr_t *Bz, fr_t *Cz);/* r1csinstance.rs multiply_vec */
void orc_eval_table_sparse(const orc_instance *, const fr_t *eq_rx /* N */, fr_t *eA, fr_t *eB, fr_t *eC /* 2V each */);
void orc_fold_top(fr_t *Z, size_t len, const fr_t *r);                                         /* bound_poly_var_top; new len = len/2 */
void orc_fold_bot(fr_t *Z, size_t len, const fr_t *r);                                         /* bound_poly_var_bot */
void orc_sc_cubic_evals(const fr_t *A, const fr_t *B, const fr_t *C, const fr_t *D, size_t len, fr_t e[3]); /* e0,e2,e3 of A*(B*C-D) */
void orc_sc_quad_evals(const fr_t *A, const fr_t *B, size_t len, fr_t e[2]);                   /* e0,e2 of A*B */
void orc_commit_rows(const fr_t *Z, size_t L, size_t R, const fr_t *blinds, const orc_mcgens *g, uint8_t *out /* L*32 */);
void orc_poly_bound(const fr_t *Z, size_t L, size_t R, const fr_t *Lvec, fr_t *out /* R */);   /* DensePolynomial::bound */
/* nizk/bullet.rs BulletReductionProof::prove, challenges given, first `rounds` rounds: (L, R) per round, folded a, b, generators */
void orc_bullet_reduce(const orc_gens *, const fr_t *a, const fr_t *b, size_t n, const fr_t *blinds /* 2 * rounds */, const fr_t *us /* rounds */,
                       size_t rounds, uint8_t *LR /* 64 * rounds */, fr_t *a_out, fr_t *b_out, uint8_t *G_out32 /* (n >> rounds) * 32 */);
void orc_gens_points(const orc_gens *, uint8_t *out /* (pc_n.n + 2) * 32: G_0..G_{R-1}, gens_1.G, h */);

/* byte-level helpers for ctypes tests */
void orc_fr_from_canon(fr_t *o, const uint8_t *b32, size_t n);   /* asserts canonical */
void orc_fr_to_canon(uint8_t *b32, const fr_t *a, size_t n);

#ifdef __cplusplus
}
#endif
#endif
