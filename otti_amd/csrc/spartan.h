// Host-side model of the Spartan NIZK objects for the MI355X proving path: instance (sparse A,B,C), generators,
// proof layout, sigma protocols, verifier.  The prover's data-parallel work lives in k_*.hip / prover.cpp.
// Mirrors upstream libspartan's public API for this path [RECALL lib.rs: Instance, VarsAssignment, InputsAssignment,
// NIZKGens, NIZK::{prove,verify}], reached from `spzk verify --nizk` [REF /root/reference/run.py:58, run.py:100].
#pragma once
#include <functional>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <memory>
#include <stdexcept>
#include "field.h"
#include "point.h"
#include "hash.h"
#include "hostgroup.h"
#include "../../include/otti_spartan.h"

namespace otti {

struct Error : std::runtime_error { int code; Error(int c, const std::string &m) : std::runtime_error(m), code(c) {} };

inline size_t ilog2(size_t n) { size_t l = 0; while (((size_t)1 << l) < n) l++; return l; }
inline size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

// ---------------------------------------------------------------------------------------------- sparse matrices
// 32-bit indices (N, 2V <= 2^31); values are Montgomery-form Fr.  The host keeps the entry list; the access paths of multiply_vec /
// compute_eval_table_sparse (CSR by row / by column) exist in HBM only (device.h DeviceCsrSet, built by k_sparse.hip).
struct SparseMat {
    std::vector<uint32_t> row, col; std::vector<Fr> val;    // entry list in caller order (as upstream's Vec<SparseMatEntry>)
};

struct DeviceInstance;   // device.h / k_sparse.hip
struct DeviceShard;      // device.h / k_sparse.hip: this rank's slice of the instance when one proof runs over several GPUs (shard.h)
struct DeviceGens;       // device.h / k_msm.hip

struct Instance {
    size_t num_cons = 0, num_vars = 0, num_inputs = 0;      // padded cons / vars (powers of two)
    size_t given_cons = 0;                                  // num_cons as passed to Instance::new (SNARK mode: upstream pads 0 / 1 constraints with explicit entries)
    SparseMat M[3];                                         // A, B, C
    std::shared_ptr<DeviceInstance> dev;                    // uploaded lazily on the first GPU prove
    std::shared_ptr<DeviceShard> shard;                     // uploaded lazily on the first sharded prove
    // Fr tuple evaluation used by the verifier: (A,B,C)(rx, ry)
    void evaluate(const std::vector<Fr> &rx, const std::vector<Fr> &ry, Fr out[3]) const;
    bool is_sat(const std::vector<Fr> &vars_padded, const std::vector<Fr> &inputs) const;
};
std::unique_ptr<Instance> instance_new(size_t num_cons, size_t num_vars, size_t num_inputs, const otti_entry *A, size_t nA,
                                       const otti_entry *B, size_t nB, const otti_entry *C, size_t nC);

// ---------------------------------------------------------------------------------------------- generators
// Every MultiCommitGens upstream derives for this path comes from ONE SHAKE256 stream (label "gens_r1cs_sat"):
// stream point j is P[j].  gens_pc.gens_n = (P[0..R), h=P[R+1]); gens_pc.gens_1 = gens_sc.gens_1 = (P[R], h=P[R+1]);
// gens_sc.gens_3 = (P[0..3), h=P[3]); gens_sc.gens_4 = (P[0..4), h=P[4]).  A commitment is therefore a short list of
// (stream index, scalar) terms over one shared fixed-base table.
struct Term { uint32_t base; Fr s; };
struct GensView { std::vector<uint32_t> G; uint32_t h; };   // indices into P
struct Gens {
    size_t num_vars_padded = 0, R = 0;
    std::vector<Pt> P;
    GensView pc_n, pc_1, sc_1, sc_3, sc_4;
    std::vector<int> small_slot;                            // P index -> slot in small_tables, or -1
    std::vector<FixedBaseTable> small_tables;               // host tables for P[0..4], P[R], P[R+1]
    std::shared_ptr<DeviceGens> dev;                        // device window table, built lazily
    Pt commit_terms(const Term *t, size_t n) const;         // host, fixed-base tables only
    PtFe commit_terms_fe(const Term *t, size_t n) const;    // the same, staying in the five-limb form of the sequential path (hostfast.h)
    void commit_terms_c(uint8_t out[32], const Term *t, size_t n) const { Pt p = commit_terms(t, n); pt_encode(out, p); }
    Pt commit_generic(const Fr *v, size_t n, const Fr &blind, const GensView &g) const;   // host MSM (verifier)
};
std::unique_ptr<Gens> gens_new(size_t num_cons, size_t num_vars, size_t num_inputs);
// sum_{i<n} s[i] * g.P[i] from g's resident device window table; false = not available (the caller computes it on the host)
typedef bool (*FixedBaseMsmHook)(const Gens &g, const Fr *s, size_t n, Pt &out);
extern FixedBaseMsmHook g_fixed_base_msm_hook;

// a DotProductProofGens inside a generator stream: gens_n.G = P[0 .. R), gens_n.h = P[h_n], gens_1 = (P[g1], h = P[h1])
struct PcView { uint32_t h_n, g1, h1; size_t R; };
std::vector<Pt> derive_generators(const char *label, size_t count);      // MultiCommitGens::new stream

std::vector<Fr> eq_evals_host(const Fr *r, size_t ell);                  // EqPolynomial::evals

// ---------------------------------------------------------------------------------------------- proof layout
typedef uint8_t Cmp[32];
struct CPoint { uint8_t b[32]; };
// The verifiers' VARIABLE-base sums  sum_i s[i] * decode(C[i])  over the sqrt(V) row commitments of a polynomial commitment
// (PolyEvalProof::verify: the points come from the proof / the computation commitment, so no table exists for them).  With a device:
// begin() uploads the compressed points and starts their decompression at once — the commitments are known long before the
// scalars, which depend on the last sum-check challenges — and finish() runs the LDS-bucket Pippenger over them (k_msm.hip
// k_msm_var) and combines the windows on the host.  finish() returns 0, or OTTI_ERR_VERIFY_DECOMPRESS when a point does not decode
// (what the host path reports); either call may decline (nullptr / negative) and the caller then does the work on the host cores.
struct RowSumJob;
typedef RowSumJob *(*RowSumBeginHook)(const CPoint *C, size_t n);
typedef int (*RowSumFinishHook)(RowSumJob *job, const Fr *s, Pt &out);   // consumes the job (also on failure); s == nullptr: just drop it
extern RowSumBeginHook g_row_sum_begin_hook;
extern RowSumFinishHook g_row_sum_finish_hook;
struct DotProductProof { CPoint delta, beta; std::vector<Fr> z; Fr z_delta, z_beta; };
struct ZKSumcheckProof { std::vector<CPoint> comm_polys, comm_evals; std::vector<DotProductProof> proofs; };
struct KnowledgeProof { CPoint alpha; Fr z1, z2; };
struct ProductProof { CPoint alpha, beta, delta; Fr z[5]; };
struct EqualityProof { CPoint alpha; Fr z; };
struct DotProductProofLog { std::vector<CPoint> L_vec, R_vec; CPoint delta, beta; Fr z1, z2; };
struct NizkProof {
    std::vector<CPoint> comm_vars;
    ZKSumcheckProof sc1;
    CPoint claims_phase2[4];                                 // Az, Bz, Cz, Az*Bz
    KnowledgeProof pok; ProductProof prod;
    EqualityProof eq1;
    ZKSumcheckProof sc2;
    CPoint comm_vars_at_ry;
    DotProductProofLog polyeval;
    EqualityProof eq2;
    std::vector<Fr> rx, ry;
    std::vector<uint8_t> serialize() const;                  // bincode layout of upstream `NIZK`
    static NizkProof parse(const uint8_t *p, size_t n);      // throws Error(OTTI_ERR_MALFORMED_PROOF)
};

// ---------------------------------------------------------------------------------------------- sigma protocols (nizk/mod.rs)
KnowledgeProof knowledge_prove(CPoint &C, const Gens &g, Transcript &tr, RandomTape &tape, const Fr &x, const Fr &r);
EqualityProof equality_prove(const Gens &g, Transcript &tr, RandomTape &tape, const Fr &v1, const Fr &s1, const Fr &v2, const Fr &s2);
ProductProof product_prove(CPoint &X, CPoint &Y, CPoint &Z, const Gens &g, Transcript &tr, RandomTape &tape, const Fr &x, const Fr &rX,
                           const Fr &y, const Fr &rY, const Fr &z, const Fr &rZ);
void unipoly_from_evals(Fr *c, const Fr *e, size_t n);
Fr unipoly_eval(const Fr *c, size_t n, const Fr &r);

// One round of ZKSumcheckInstanceProof::prove_{quad,cubic_with_additive_term} after the table sums are known.
// The prover's randomness (RandomTape) is a separate transcript that depends only on the seed, so everything derived from it
// alone is drawn and committed for all rounds up front (RoundPre; one batched fixed-base MSM on the device): the per-round
// host work on the sequential Fiat-Shamir path is then 4+1+2+1 fixed-base terms instead of 5+2+2+5+2.
struct RoundPre {
    Fr d[4], r_delta, r_beta;          // DotProductProof::prove draws: d_vec, r_delta, r_beta
    Pt delta; CPoint delta_c;          // commit(d_vec, r_delta) over gens_n, and its compressed form
    Pt bp_h, be_h, rb_h;               // blinds_poly[j]*h_n, blinds_evals[j]*h_1, r_beta*h_1
    PtFe bp_fe, be_fe, rb_fe;          // the same in the host's five-limb form (RoundPre::to_fe)
    void to_fe() { bp_fe = ptfe_from(bp_h); be_fe = ptfe_from(be_h); rb_fe = ptfe_from(rb_h); }
};
struct SumcheckState { Fr claim; CPoint comm_claim; Fr blind_claim; std::vector<Fr> blinds_poly, blinds_evals; std::vector<RoundPre> pre; };
struct RoundPart1 { Fr poly[4]; size_t ne; Fr r_j; };
// draws blinds_poly, blinds_evals and every round's (d_vec, r_delta, r_beta) from the tape in upstream's order
void sumcheck_draw_tape(SumcheckState &st, ScalarSource &tape, size_t num_rounds, size_t ne);
RoundPart1 sumcheck_round_begin(ZKSumcheckProof &pf, size_t j, const Fr *evals, size_t ne, const SumcheckState &st, const Gens &g,
                                const GensView &gn, Transcript &tr);
void sumcheck_round_finish(ZKSumcheckProof &pf, size_t j, const RoundPart1 &p1, SumcheckState &st, const Gens &g, const GensView &gn,
                           Transcript &tr);

// ---------------------------------------------------------------------------------------------- verifier (lib.rs NIZK::verify)
// inst_evals: optional (A,B,C)(rx,ry) computed elsewhere (the device: the O(nnz + N + V) part of verification); NULL => host
// fetch: the values are on their way (an evaluation begun on the device before the call: device.h instance_evaluate_begin) and are
// collected through it at the point where the verifier first needs them, after both sum-checks
typedef std::function<void(Fr out[3])> InstEvalFetch;
int nizk_verify(const Instance &inst, const std::vector<Fr> &inputs, const Gens &g, const void *tlabel, size_t tlabel_len,
                const uint8_t *proof, size_t proof_len, const Fr *inst_evals = nullptr, const InstEvalFetch *fetch = nullptr);

// ---------------------------------------------------------------------------------------------- prover (prover.cpp; GPU)
struct ProveTimings { double ms[8]; };   // polycommit, multiply_vec, sc_phase_one, eval_table_sparse, sc_phase_two, polyeval, total, (spare)
std::vector<uint8_t> nizk_prove_gpu(Instance &inst, const std::vector<Fr> &vars_padded, const std::vector<Fr> &inputs, Gens &g,
                                    const void *tlabel, size_t tlabel_len, const uint8_t *seed32, ProveTimings *tm);

// synthetic satisfiable instance of SURVEY.md section 8(d)
void synth_r1cs(size_t n, size_t num_inputs, uint64_t seed, std::vector<otti_entry> &A, std::vector<otti_entry> &B,
                std::vector<otti_entry> &C, std::vector<uint8_t> &vars32, std::vector<uint8_t> &inputs32);

void synth_r1cs_compiler_like(size_t n, size_t num_inputs, uint64_t seed, std::vector<otti_entry> &A, std::vector<otti_entry> &B,
                              std::vector<otti_entry> &C, std::vector<uint8_t> &vars32, std::vector<uint8_t> &inputs32);

}  // namespace otti
