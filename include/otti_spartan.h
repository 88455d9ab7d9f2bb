/*
 * libottispartan — C ABI of the MI355X-native Spartan NIZK proving path for Otti.
 *
 * Drop-in boundary.  The reference reaches this path in two ways:
 *   - as a process: `spzk verify --nizk X.zkif X.inp.zkif X.wit.zkif`        [REF /root/reference/run.py:52-59, run.py:96-100]
 *   - in-process, as libspartan calls from rust-circ `--action spartan`       [REF /root/reference/run.py:147]
 * The library-level interface those callers bind is upstream libspartan's `src/lib.rs` [RECALL — the Spartan/ submodule is an
 * empty directory in the reference mount, /root/reference/.gitmodules:4-6]; every entry point below names the item it replaces.
 *
 * Conventions: plain pointers and sizes; int32 status (0 = ok, negatives mirror upstream's error enums); inputs are borrowed for
 * the duration of the call; outputs are owned by the library until the matching *_free; no exceptions cross the ABI.
 * Threads: creating, preparing and freeing a handle is single-threaded; once otti_prepare_device has run, instance, generator and
 * witness handles are read-only and any number of threads may PROVE (otti_nizk_prove, otti_nizk_prove_resident) and verify with
 * them concurrently — every calling thread gets its own device context (HIP stream, result mailbox, HBM workspace, helper threads),
 * so their proofs overlap on the GPU; a single proof is a chain of sequential rounds that leaves most of the chip idle.  (Upstream's
 * prover is likewise re-entrant: `NIZK::prove(&inst, .., &gens, ..)` borrows instance and generators immutably.)  All scalars are 32-byte canonical little-endian encodings of GF(l),
 * l = 2^252 + 27742317777372353535851937790883648493, unless a comment says "Montgomery" (the in-HBM layout: value*2^256 mod l).
 */
#ifndef OTTI_SPARTAN_H
#define OTTI_SPARTAN_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* upstream R1CSError / ProofVerifyError [RECALL src/errors.rs] */
enum {
    OTTI_OK = 0,
    OTTI_ERR_NON_POW2_CONS = -1,
    OTTI_ERR_NON_POW2_VARS = -2,
    OTTI_ERR_INVALID_NUM_INPUTS = -3,
    OTTI_ERR_INVALID_NUM_VARS = -4,
    OTTI_ERR_INVALID_SCALAR = -5,
    OTTI_ERR_INVALID_INDEX = -6,
    OTTI_ERR_VERIFY_INTERNAL = -10,       /* ProofVerifyError::InternalError */
    OTTI_ERR_VERIFY_DECOMPRESS = -11,     /* ProofVerifyError::DecompressionError */
    OTTI_ERR_MALFORMED_PROOF = -12,       /* bincode deserialisation failure */
    OTTI_ERR_NO_DEVICE = -20,             /* no gfx950 device / HIP failure: the proving path has no CPU fallback */
    OTTI_ERR_BAD_ARG = -21,
    OTTI_ERR_IO = -22,                    /* zkInterface file unreadable / malformed */
    OTTI_ERR_INTERNAL = -23
};

/* one non-zero of A, B or C: upstream `(usize, usize, [u8; 32])` tuples passed to Instance::new */
typedef struct { uint64_t row; uint64_t col; uint8_t val[32]; } otti_entry;

typedef struct otti_instance otti_instance;   /* upstream `Instance` */
typedef struct otti_gens otti_gens;           /* upstream `NIZKGens` */

/* flags for otti_nizk_prove */
#define OTTI_FLAG_GPU 0x1u                    /* required: the only proving backend */

/* Instance::new(num_cons, num_vars, num_inputs, &A, &B, &C) -> Result<Instance, R1CSError>.
   Pads cons/vars to powers of two, shifts columns >= num_vars by the padding, rejects bad indices / non-canonical scalars. */
int32_t otti_instance_new(uint64_t num_cons, uint64_t num_vars, uint64_t num_inputs,
                          const otti_entry *A, size_t nA, const otti_entry *B, size_t nB, const otti_entry *C, size_t nC,
                          otti_instance **out);
void    otti_instance_free(otti_instance *inst);
/* padded sizes as seen by the prover (Instance.inst.get_num_cons / get_num_vars / get_num_inputs) */
int32_t otti_instance_dims(const otti_instance *inst, uint64_t *num_cons, uint64_t *num_vars, uint64_t *num_inputs);
/* Instance::is_sat(&vars, &inputs) -> Result<bool, R1CSError>  (host check, used by spzk before proving) */
int32_t otti_instance_is_sat(const otti_instance *inst, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs,
                             int32_t *sat);

/* NIZKGens::new(num_cons, num_vars, num_inputs) */
int32_t otti_gens_new(uint64_t num_cons, uint64_t num_vars, uint64_t num_inputs, otti_gens **out);
void    otti_gens_free(otti_gens *gens);
/* compressed generator stream P[0 .. count) (tests pin it against SURVEY App. B) */
int32_t otti_gens_points(const otti_gens *gens, uint8_t *out32, size_t count);
/* the device-side fixed-base window table of these generators, once built (otti_prepare_device or the first proof): window width c
   (OTTI_MSM_WINDOW pins it; otherwise the widest whose table fits OTTI_MSM_TABLE_GB, default 128) and its size; zeros before that.
   No reference counterpart: dalek's vartime MSM builds per-call tables. */
int32_t otti_gens_table_info(const otti_gens *gens, uint32_t *window_bits, uint64_t *table_bytes);
/* frees the device-side window table (tens of GB); it is built again by the next otti_prepare_device or proof with these generators.
 * For a process that moves between instance sizes: two wide tables do not fit one card.  Not while a proof with them is running. */
int32_t otti_gens_release_device(otti_gens *gens);
/* what building the table took, in ms: the HBM allocations (driver work: differs widely between boxes and between a fresh and a used
 * process) and the upload + fill kernels (proportional to the table); zeros before it has been built */
int32_t otti_gens_build_ms(const otti_gens *gens, double *alloc_ms, double *kernels_ms);

/* NIZK::prove(&inst, vars, &inputs, &gens, &mut Transcript::new(tlabel)) -> NIZK, bincode-serialised.
   VarsAssignment::new / InputsAssignment::new validation (InvalidScalar) happens here.
   seed32: 32 bytes seeding the prover's RandomTape (upstream uses OsRng; NULL => OS entropy).
   stage_ms: optional 8 doubles — polycommit, multiply_vec, sc_phase_one, eval_table_sparse, sc_phase_two, polyeval, total, 0. */
int32_t otti_nizk_prove(otti_instance *inst, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs,
                        otti_gens *gens, const uint8_t *tlabel, size_t tlabel_len, const uint8_t *seed32, uint32_t flags,
                        uint8_t **proof, size_t *proof_len, double *stage_ms);
/* The same proof with the witness already resident in HBM (z = vars || 1 || inputs || 0.. in Montgomery form): upload once,
   prove many times.  This is the boundary bench.py times: no PCIe traffic inside otti_nizk_prove_resident except the
   per-round field elements and the compressed points of the proof itself. */
typedef struct otti_witness otti_witness;
int32_t otti_witness_upload(otti_instance *inst, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs, otti_witness **out);
void    otti_witness_free(otti_witness *w);
int32_t otti_nizk_prove_resident(otti_instance *inst, otti_witness *wit, otti_gens *gens, const uint8_t *tlabel, size_t tlabel_len,
                                 const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms);
/* ---- one proof over several GPUs of one node (no reference counterpart: upstream `spzk` is one process
   [REF /root/reference/run.py:52-59]).  One process per GPU.  Every process calls otti_shard_init with the same segment name (a
   fresh, unique name per job: it names a POSIX shared-memory object, removed again once all ranks are attached), its rank and the
   world size (a power of two, at most the number of witness-matrix rows); then each calls otti_nizk_prove_sharded with the SAME
   instance, witness, generators, label and seed.  Every rank returns the same proof bytes, identical to otti_nizk_prove_resident's.
   While proving, ranks exchange only what the host has to hash anyway (per-round sums, row commitments, the partial L^T Z vector)
   through that segment; the witness is replicated by the caller beforehand (e.g. torch.distributed broadcast over RCCL).
   otti_shard_allgather / otti_shard_allreduce are the exchange primitives themselves (host memory; usable without a GPU). */
int32_t otti_shard_init(const char *segment_name, uint32_t rank, uint32_t world);
int32_t otti_shard_finalize(void);
/* transport of the per-round sums: 0 = node-local mailbox (default), 1 = RCCL ncclAllReduce(ncclUint64, ncclSum) over u64 lanes
   (environment OTTI_SHARD_TRANSPORT=rccl at otti_shard_init; needs the GPU) */
int32_t otti_shard_info(uint32_t *rank, uint32_t *world, uint32_t *transport);
int32_t otti_shard_allgather(const void *mine, size_t nbytes, void *out /* world * nbytes */);
int32_t otti_shard_allreduce(uint8_t *scalars32 /* canonical, in place */, size_t count);
int32_t otti_nizk_prove_sharded(otti_instance *inst, otti_witness *wit, otti_gens *gens, const uint8_t *tlabel, size_t tlabel_len,
                                const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms);
/* NIZK::verify(&self, &inst, &inputs, &mut Transcript::new(tlabel), &gens) -> Result<(), ProofVerifyError> */
int32_t otti_nizk_verify(const otti_instance *inst, const uint8_t *inputs32, size_t ninputs, const otti_gens *gens,
                         const uint8_t *tlabel, size_t tlabel_len, const uint8_t *proof, size_t proof_len);
/* ---- SNARK mode (`spzk verify` without --nizk): upstream lib.rs SNARKGens / ComputationCommitment / SNARK [RECALL].
   otti_snark_encode = SNARK::encode (once per circuit: the commitment to A, B, C — on the GPU); otti_snark_prove = SNARK::prove
   (R1CSProof as in NIZK mode + R1CSEvalProof: memory-checking product circuits, batched cubic sum-checks, three polynomial
   evaluation proofs); otti_snark_verify = SNARK::verify, which needs the commitment only (host; sub-linear in the circuit).
   num_nz_entries: the largest number of non-zeros of A, B, C (what spartan-zkinterface passes to SNARKGens::new).
   stage_ms (otti_snark_prove): 10 doubles — the six R1CSProof stages of otti_nizk_prove, [6] derefs commitment, [7] product
   circuits, [8] hash layer, [9] total. */
typedef struct otti_snark_gens otti_snark_gens;      /* upstream `SNARKGens` */
typedef struct otti_comp_comm otti_comp_comm;        /* upstream `ComputationCommitment` (+ `ComputationDecommitment` when made by encode) */
int32_t otti_snark_gens_new(uint64_t num_cons, uint64_t num_vars, uint64_t num_inputs, uint64_t num_nz_entries, otti_snark_gens **out);
void    otti_snark_gens_free(otti_snark_gens *gens);
int32_t otti_snark_encode(otti_instance *inst, otti_snark_gens *gens, otti_comp_comm **out);
int32_t otti_comp_comm_bytes(const otti_comp_comm *comm, uint8_t **out, size_t *len);        /* bincode of the commitment; free with otti_buf_free */
int32_t otti_comp_comm_from_bytes(const uint8_t *buf, size_t len, otti_comp_comm **out);     /* the verifier's copy (no decommitment) */
void    otti_comp_comm_free(otti_comp_comm *comm);
int32_t otti_snark_prove(otti_instance *inst, otti_comp_comm *comm, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs,
                         otti_snark_gens *gens, const uint8_t *tlabel, size_t tlabel_len, const uint8_t *seed32, uint32_t flags,
                         uint8_t **proof, size_t *proof_len, double *stage_ms);
/* the same with the assignment already resident in HBM (otti_witness_upload): what a long-lived prover calls per proof */
int32_t otti_snark_prove_resident(otti_instance *inst, otti_comp_comm *comm, otti_witness *wit, otti_snark_gens *gens, const uint8_t *tlabel,
                                  size_t tlabel_len, const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms);
/* this rank's part of ONE SNARK::prove over the GPUs of a node (after otti_shard_init, like otti_nizk_prove_sharded: every rank passes the
   same instance, commitment, resident witness, generators, label and seed and receives the same proof bytes): the R1CS satisfiability proof
   sharded as in NIZK mode, the rows of the derefs commitment dealt out over the ranks, the product circuits split by residue classes of the
   element index (their per-round sums cross the ranks); host rounds and evaluation proofs run on every rank alike */
int32_t otti_snark_prove_sharded(otti_instance *inst, otti_comp_comm *comm, otti_witness *wit, otti_snark_gens *gens, const uint8_t *tlabel,
                                 size_t tlabel_len, const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms);
int32_t otti_snark_verify(const otti_comp_comm *comm, const uint8_t *inputs32, size_t ninputs, const otti_snark_gens *gens,
                          const uint8_t *tlabel, size_t tlabel_len, const uint8_t *proof, size_t proof_len);
void    otti_buf_free(void *p);
/* copies the calling thread's last error message (NUL-terminated, truncated to cap) */
size_t  otti_last_error(char *buf, size_t cap);

/* upload instance / build the generator window table ahead of the first prove (both are otherwise lazy); with both NULL: only bring
   the HIP runtime and the device context up (a one-shot caller does this on a side thread while it parses its input) */
int32_t otti_prepare_device(otti_instance *inst, otti_gens *gens);
/* number of visible gfx950 devices (0 when none; never initialises a context) */
int32_t otti_device_count(void);
/* Self-test of the host-side fast paths that sit on the prover's sequential Fiat-Shamir path (five-limb GF(2^255-19): point
   compression, fixed-base window tables) against the generic field code and a variable-base multiplication, on `iterations`
   pseudo-random inputs.  0 = consistent.  Needs no GPU; the prover itself only runs on one, so this is how the CPU test suite
   reaches that code. */
int32_t otti_host_selftest(uint32_t iterations);
/* Measurement aid: nanoseconds per operation of the host primitives on the provers' sequential path, on the calling machine:
 * [0] fixed-base scalar multiplication (8-bit windows), [1] ristretto compression, [2] Keccak-f[1600], [3] transcript append of a point
 * + challenge scalar, [4] GF(l) multiplication, [5] GF(l) inversion, [6] hand-off of an empty task to a helper thread and back,
 * [7] / [8] the two halves of one zero-knowledge sum-check round's host work (up to the challenge / after it), [9] threads used. */
int32_t otti_host_microbench(double out[10]);
/* Measurement aid, no GPU: microseconds per LAYER of the host's last sum-check rounds in SNARK mode (hosttail.h: every round's sums + fold
 * over np product instances and nd triples with tables of T elements, `threads` threads, mean of `reps` layers on random tables):
 * out[0] with the AVX-512 IFMA form (0 when the CPU lacks the instructions or OTTI_HOST_FR8=0), out[1] with the scalar form. */
int32_t otti_host_tail_bench(uint32_t np, uint32_t nd, uint64_t T, uint32_t threads, uint32_t reps, double out[2]);

/* ---- zkInterface ingest (replaces spartan-zkinterface's reader; schema zkinterface 1.x) ---- */
typedef struct {
    uint64_t num_cons, num_vars, num_inputs;
    otti_entry *A, *B, *C; size_t nA, nB, nC;
    uint8_t *vars32; size_t nvars;         /* witness assignment, canonical LE */
    uint8_t *inputs32; size_t ninputs;     /* instance (public input) assignment */
} otti_r1cs;
int32_t otti_zkif_load(const char *circuit_path, const char *inputs_path, const char *witness_path, otti_r1cs **out);
/* writes the three-file split the reference compiler produces [REF run.py:47-49] from an R1CS in z-order [vars | 1 | inputs] */
int32_t otti_zkif_write(const otti_r1cs *r, const char *circuit_path, const char *inputs_path, const char *witness_path);
void    otti_r1cs_free(otti_r1cs *r);
/* synthetic satisfiable R1CS: num_cons = num_vars = n, one non-zero per row per matrix */
int32_t otti_synth_r1cs(uint64_t n, uint64_t num_inputs, uint64_t seed, otti_r1cs **out);
/* second distribution of SURVEY 8d ("compiler-like"): 90 % of the witness < 2^64, 1..8 non-zeros per row, small signed coefficients,
   a heavily used constant column, one 300-entry row */
int32_t otti_synth_r1cs_compiler_like(uint64_t n, uint64_t num_inputs, uint64_t seed, otti_r1cs **out);

/* ---- kernel-level entry points (tests / bench).  h_* = host pointers; elements are 32-byte Montgomery-form Fr.
        Each call stages inputs to HBM, runs the named kernel(s) on the library's stream, and copies results back;
        kernel_ms (optional) receives the HIP-event time of the kernel launches alone. ---- */
/* Fr/Fp/point self-test kernels: out[i] = a[i] * b[i] etc.  op: 0 mul, 1 add, 2 sub */
int32_t otti_k_fr_op(int32_t op, const uint8_t *h_a, const uint8_t *h_b, uint8_t *h_out, size_t n, float *kernel_ms);
/* canonical LE <-> Montgomery on the device */
int32_t otti_k_fr_from_canonical(const uint8_t *h_in, uint8_t *h_out, size_t n);
int32_t otti_k_fr_to_canonical(const uint8_t *h_in, uint8_t *h_out, size_t n);
/* R1CSInstance::multiply_vec: z has 2*num_vars entries; outputs num_cons entries each */
int32_t otti_k_multiply_vec(otti_instance *inst, const uint8_t *h_z, uint8_t *h_Az, uint8_t *h_Bz, uint8_t *h_Cz, float *kernel_ms);
/* compute_eval_table_sparse x3 fused with r_A*A + r_B*B + r_C*C: eq_rx has num_cons entries, out 2*num_vars */
int32_t otti_k_eval_table_sparse(otti_instance *inst, const uint8_t *h_eq_rx, const uint8_t *h_rABC /* 3 */, uint8_t *h_out, float *kernel_ms);
/* EqPolynomial::evals */
int32_t otti_k_eq_evals(const uint8_t *h_r, size_t ell, uint8_t *h_out, float *kernel_ms);
/* DensePolynomial::bound_poly_var_top / _bot (in place on a staged copy; out has len/2 entries) */
int32_t otti_k_fold_top(const uint8_t *h_Z, size_t len, const uint8_t *h_r, uint8_t *h_out, float *kernel_ms);
int32_t otti_k_fold_bot(const uint8_t *h_Z, size_t len, const uint8_t *h_r, uint8_t *h_out, float *kernel_ms);
/* one round of prove_cubic_with_additive_term's table work: (e0,e2,e3) of A*(B*C-D) over tables of length len */
int32_t otti_k_sc_cubic_round(const uint8_t *h_A, const uint8_t *h_B, const uint8_t *h_C, const uint8_t *h_D, size_t len, uint8_t *h_e3, float *kernel_ms);
/* fused fold(r) + next round's sums: tables of length len are folded to len/2 (written to h_out4, 4*len/2) and (e0,e2,e3) returned */
int32_t otti_k_sc_cubic_fold_round(const uint8_t *h_A, const uint8_t *h_B, const uint8_t *h_C, const uint8_t *h_D, size_t len,
                                   const uint8_t *h_r, uint8_t *h_out4, uint8_t *h_e3, float *kernel_ms);
/* one round of prove_quad: (e0,e2) of A*B */
int32_t otti_k_sc_quad_round(const uint8_t *h_A, const uint8_t *h_B, size_t len, uint8_t *h_e2, float *kernel_ms);
int32_t otti_k_sc_quad_fold_round(const uint8_t *h_A, const uint8_t *h_B, size_t len, const uint8_t *h_r, uint8_t *h_out2, uint8_t *h_e2, float *kernel_ms);
/* DensePolynomial::commit_inner: L rows of R scalars -> L compressed points C_i = sum_j Z[iR+j] P[j] + blinds[i] P[R+1] */
/* Self-test of the "armed" launches the provers use for their small sequential rounds (a kernel queued before its challenge is known
 * and released through pinned host memory): the quadratic fold + sums round on the caller's tables (length len, a power of two >= 8),
 * plain, armed + released after hold_us microseconds, and armed + aborted.  out2 (len/2 + len/2 elements) and e2 (2 elements) are the
 * plain launch's results; any disagreement between the three ways is OTTI_ERR_INTERNAL. */
int32_t otti_k_armed_selftest(const uint8_t *A, const uint8_t *B, size_t len, const uint8_t *r, uint32_t hold_us, uint8_t *out2, uint8_t *e2);
int32_t otti_k_msm_rows(otti_gens *gens, const uint8_t *h_Z, size_t L, size_t R, const uint8_t *h_blinds, uint8_t *h_out32, float *kernel_ms);
/* PolyEvalProof::verify's C_LZ on the device [dense_mlpoly.rs PolyEvalProof::verify -> GroupElement::vartime_multiscalar_mul, RECALL]:
   out = compress(sum_i s[i] * decompress(C[i])), n >= 256 compressed ristretto255 points, scalars in Montgomery form.  Batch decompression,
   LDS-bucket Pippenger, window recombination on the host.  OTTI_ERR_VERIFY_DECOMPRESS if an encoding does not decode. */
int32_t otti_k_row_sum(const uint8_t *h_compressed32, size_t n, const uint8_t *h_scalars_mont32, uint8_t *h_out32);

/* ---- the kernels the prover actually launches for phase one, the evaluation proof and the bullet reduction (host pointers, as above).
        No reference counterpart beyond the upstream functions named; these exist so that a failing proof points at a kernel. ---- */
/* eq "pyramid": level k (eq over the LAST k of the n variables, 2^k entries) at out + 2^k - 1; n <= 13; out has 2^(n+1) - 1 entries */
int32_t otti_k_eq_pyramid(const uint8_t *h_r, size_t n, uint8_t *h_out);
/* prove_cubic_with_additive_term with the eq table factored out: S_t = sum_i E[i] (B_t C_t - D_t)[i], t = 0, 2, 3, over tables of
   length len, E = EqPolynomial(h_tau[0 .. log2(len) - 1)).evals() supplied as the challenge suffix it is built from */
int32_t otti_k_sc_cubic3_round(const uint8_t *h_B, const uint8_t *h_C, const uint8_t *h_D, size_t len, const uint8_t *h_tau, uint8_t *h_e3, float *kernel_ms);
/* fused bound_poly_var_top(r) + the next round's S_t: tables of length len (>= 4) fold to len/2 (h_out3: 3 * len/2 elements);
   h_tau holds the log2(len) - 2 variables E of the folded round is built from */
int32_t otti_k_sc_cubic3_fold_round(const uint8_t *h_B, const uint8_t *h_C, const uint8_t *h_D, size_t len, const uint8_t *h_r, const uint8_t *h_tau,
                                    uint8_t *h_out3, uint8_t *h_e3, float *kernel_ms);
/* DensePolynomial::bound: out[j] = sum_i Lv[i] * Z[i*R + j] */
int32_t otti_k_poly_bound(const uint8_t *h_Z, size_t L, size_t R, const uint8_t *h_Lv, uint8_t *h_out, float *kernel_ms);
/* One BulletReductionProof::prove round on the ORIGINAL generators (k_msm.hip): state (a, b: 2*n_cur elements if fold else n_cur; s: R
   coefficients of the original generators).  If fold, the previous challenge (u, u^-1) is applied first.  Returns compressed L, R
   (64 bytes; blinds h_blinds2 = {blind_L, blind_R}) and the state the next round starts from (a, b: n_cur elements; s: R). */
int32_t otti_k_bullet_round(otti_gens *gens, size_t n_cur, int32_t fold, const uint8_t *h_u, const uint8_t *h_uinv, const uint8_t *h_a, const uint8_t *h_b,
                            const uint8_t *h_s, const uint8_t *h_blinds2, uint8_t *h_a_out, uint8_t *h_b_out, uint8_t *h_s_out, uint8_t *h_LR64, float *kernel_ms);
/* the closing fold of the reduction (length 2 -> 1): a, b of 2 elements and s of R are folded in place by (u, u^-1) */
int32_t otti_k_bullet_last_fold(size_t R, const uint8_t *h_u, const uint8_t *h_uinv, uint8_t *h_a2, uint8_t *h_b2, uint8_t *h_s);

/* ---- the same kernels on DEVICE pointers and a caller-chosen HIP stream ("host or device pointers + a stream
        handle"): nothing is staged through PCIe, so a kernel can be benchmarked or composed from outside the library.
        stream: a hipStream_t passed as void* (NULL = the calling thread's library stream).  d_* = device memory holding 32-byte
        Montgomery-form elements; h_* = small host arrays (challenges come from the transcript).  Calls that return round sums
        (h_e*) wait for that launch's result; the others only enqueue. ---- */
int32_t otti_kd_multiply_vec(otti_instance *inst, const void *d_z, void *d_Az, void *d_Bz, void *d_Cz, void *stream);
int32_t otti_kd_eval_table_sparse(otti_instance *inst, const void *d_eq_rx, const uint8_t *h_rABC, void *d_out, void *stream);
int32_t otti_kd_eq_evals(const uint8_t *h_r, size_t ell, void *d_out, void *stream);
int32_t otti_kd_fold_top(void *d_Z, size_t len, const uint8_t *h_r, void *stream);                    /* in place: len -> len/2 */
int32_t otti_kd_fold_bot(const void *d_Z, void *d_out, size_t len, const uint8_t *h_r, void *stream);
int32_t otti_kd_sc_cubic_round(const void *d_A, const void *d_B, const void *d_C, const void *d_D, size_t len, uint8_t *h_e3, void *stream);
int32_t otti_kd_sc_cubic_fold_round(void *d_A, void *d_B, void *d_C, void *d_D, size_t len, const uint8_t *h_r, uint8_t *h_e3, void *stream);   /* in place */
int32_t otti_kd_sc_quad_round(const void *d_A, const void *d_B, size_t len, uint8_t *h_e2, void *stream);
int32_t otti_kd_sc_quad_fold_round(void *d_A, void *d_B, size_t len, const uint8_t *h_r, uint8_t *h_e2, void *stream);                            /* in place */
/* compressed row commitments land in d_out32 (32 * L bytes of device memory) */
int32_t otti_kd_msm_rows(otti_gens *gens, const void *d_Z, size_t L, size_t R, const void *d_blinds, void *d_out32, void *stream);
/* plain device memory for callers without a HIP runtime of their own (ctypes tests); hipMalloc'ed buffers of any origin work as well */
int32_t otti_dev_alloc(size_t nbytes, void **d_out);
int32_t otti_dev_free(void *d);
int32_t otti_dev_upload(void *d_dst, const void *h_src, size_t nbytes);
int32_t otti_dev_download(void *h_dst, const void *d_src, size_t nbytes);
/* a HIP stream of the runtime the library itself is linked to (a caller with its own HIP code passes its hipStream_t instead) */
int32_t otti_dev_stream_create(void **stream_out);
int32_t otti_dev_stream_sync(void *stream);
int32_t otti_dev_stream_destroy(void *stream);

/* the integer-ALU roof the bulk MSM is priced against: whole-chip throughput of its mixed point addition (operands in registers,
   every CU busy), measured now (about 10 ms of GPU time) */
int32_t otti_bench_madd_peak(double *madds_per_second);
/* the second roof of the streaming kernels (sum-check rounds, sparse products, eq tables): whole-chip throughput of the Montgomery
   product in GF(l), operands in registers, measured now (about 10 ms of GPU time) */
int32_t otti_bench_fr_mul_peak(double *products_per_second);

/* per-kernel-class timing with HIP events recorded on the library's own stream around every launch of that class.
   classes: msm_rows (>= 2^16 scalars per launch: the witness commitment) msm_small msm_finish sc_cubic sc_quad spmv eq reduce poly_bound bullet other.  enable(1) also resets the counters. */
int32_t otti_stats_enable(int32_t on);
/* restrict timing to one class (call after enable): two event records per launch are not free on the latency-bound round loop */
int32_t otti_stats_select(const char *kernel_class);
int32_t otti_stats_read(const char *kernel_class, uint64_t *count, double *total_ms);
/* SNARK mode adds the classes pc_round (rounds of the batched product-circuit sum-checks) prod_layer hash_layer gather dot_many. */
/* 1 when the calling thread's next proof would use armed launches (kernels queued ahead of their challenge): off under OTTI_ARMED=0,
   while a class with armed kernels (msm_small sc_cubic sc_quad pc_round) is being timed, and with several proofs in flight */
int32_t otti_armed_launches_on(int32_t *on);

/* ---- multi-GPU plumbing: sum-check partial sums travel as 8 x u32 limbs widened to u64 lanes so that a plain integer
        sum all-reduce (RCCL ncclSum/ncclUint64, or gloo in CPU tests) followed by one normalisation gives the Fr sum ---- */
void    otti_lanes_pack(const uint8_t *fr_mont32, size_t n, uint64_t *lanes /* 8n */);
void    otti_lanes_unpack(const uint64_t *lanes, size_t n, uint8_t *fr_mont32);   /* reduces each 8-lane group mod l */

#ifdef __cplusplus
}
#endif
#endif
