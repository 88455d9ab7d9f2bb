// SNARK mode of the proving path: libspartan's computation commitment (SNARK::encode), R1CSEvalProof and SNARK::{prove, verify}
// [RECALL upstream src/lib.rs (SNARKGens, SNARK), src/r1csinstance.rs (R1CSCommitment, R1CSEvalProof), src/sparse_mlpoly.rs,
// src/product_tree.rs, src/sumcheck.rs (SumcheckInstanceProof::prove_cubic_batched); /root/reference/Spartan is an empty submodule].
// Reached from `spzk verify <files>` WITHOUT --nizk (the reference's run.py passes --nizk, /root/reference/run.py:58,100; BASELINE.json's
// metric names the SNARK).  Host objects and the verifier live in snark_host.cpp, the GPU prover in snark_prover.cpp, its kernels in
// k_snark.hip; the R1CS satisfiability proof inside it is the same device code as NIZK mode (prover.cpp r1cs_prove_device).
#pragma once
#include "spartan.h"
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <algorithm>
#include <vector>
#include <system_error>
#include <functional>
#include <thread>
#include <atomic>

namespace otti {

struct DeviceDecomm;                                         // snark_prover.cpp: the dense representation resident in HBM

// lib.rs SNARKGens: gens_r1cs_sat (the NIZK generators) + gens_r1cs_eval = SparseMatPolyCommitmentGens {gens_ops, gens_mem, gens_derefs},
// three PolyCommitmentGens over ONE SHAKE256 stream (label "gens_r1cs_eval"): a set for m variables is gens_n = P[0 .. R), gens_1.G = P[R],
// h = P[R + 1] with R = 2^(m - m/2).  `eval` holds that stream (and its device window table); the views are stream indices.
struct PcSet { size_t num_vars = 0, L = 0, R = 0; uint32_t h_n = 0, g1 = 0, h1 = 0; };
struct SnarkGens {
    std::unique_ptr<Gens> sat, eval;
    PcSet ops, mem, derefs;
    size_t num_cons = 0, num_vars = 0, num_inputs = 0;       // padded, as the instance will have them
};
std::unique_ptr<SnarkGens> snark_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs, size_t num_nz_entries);

// lib.rs ComputationCommitment (+ ComputationDecommitment when made by encode)
struct CompComm {
    size_t num_cons = 0, num_vars = 0, num_inputs = 0;       // R1CSCommitment
    size_t batch_size = 3, num_ops = 0, num_mem_cells = 0;   // SparseMatPolyCommitment
    std::vector<CPoint> comm_ops, comm_mem;
    std::shared_ptr<DeviceDecomm> dec;                       // prover side only
    std::vector<uint8_t> serialize() const;                  // bincode
    static std::unique_ptr<CompComm> parse(const uint8_t *p, size_t n);
};

// sumcheck.rs SumcheckInstanceProof (CompressedUniPoly per round: c0, c2, c3) / product_tree.rs proofs
struct LayerProofBatched { std::vector<Fr> coeffs; std::vector<Fr> left, right; };     // coeffs: 3 per round
struct ProductCircuitEvalProofBatched { std::vector<LayerProofBatched> layers; std::vector<Fr> dotp_left, dotp_right, dotp_weight; };
struct Evals4 { Fr init, audit; Fr read[3], write[3]; };
// r1csinstance.rs R1CSEvalProof = sparse_mlpoly.rs SparseMatPolyEvalProof
struct EvalProof {
    std::vector<CPoint> comm_derefs;
    Evals4 eval_row, eval_col; Fr dotp_left[3], dotp_right[3];
    ProductCircuitEvalProofBatched proof_mem, proof_ops;
    Fr h_row_addr[3], h_row_read_ts[3], h_row_audit, h_col_addr[3], h_col_read_ts[3], h_col_audit, h_val[3], h_deref_row[3], h_deref_col[3];
    DotProductProofLog pe_ops, pe_mem, pe_derefs;
};
struct SnarkProof {
    NizkProof r1cs;                                          // R1CSProof (its rx, ry are not serialised in SNARK mode)
    Fr inst_evals[3];
    EvalProof eval;
    std::vector<uint8_t> serialize() const;
    static SnarkProof parse(const uint8_t *p, size_t n);     // throws Error(OTTI_ERR_MALFORMED_PROOF)
};

void snark_append_comm(Transcript &tr, const CompComm &c);  // R1CSCommitment::append_to_transcript
// host verifier (snark_host.cpp): SNARK::verify
int snark_verify(const CompComm &comm, const std::vector<Fr> &inputs, const SnarkGens &g, const void *tlabel, size_t tlabel_len, const uint8_t *proof, size_t proof_len);
// R1CSProof::verify shared with NIZK mode (spartan_host.cpp): returns 0 and the challenges
int r1cs_verify_host(const NizkProof &P, size_t N, size_t V, const std::vector<Fr> &inputs, const Fr inst_evals[3], const Gens &g, Transcript &tr,
                     std::vector<Fr> &rx, std::vector<Fr> &ry, const InstEvalFetch *fetch = nullptr);

// verifier building blocks shared with NIZK mode (spartan_host.cpp)
struct VerifyFail { int code; };
Pt dec(const CPoint &c);                                     // throws VerifyFail{OTTI_ERR_VERIFY_DECOMPRESS}
void require(bool ok);                                       // throws VerifyFail{OTTI_ERR_VERIFY_INTERNAL}
Pt host_msm_wide(const Fr *sc, const Pt *pts, size_t n);
// sum_i s[i] * decode(C[i]) over the row commitments of a polynomial commitment (PolyEvalProof::verify's C_LZ).  Construct it as soon as
// the commitments are known (with a device their decompression starts right away, spartan.h RowSumBeginHook), call finish() when the
// scalars are; without a device, or when it declines, both steps run on the host cores inside finish().  Throws VerifyFail.
struct RowSum {
    RowSumJob *job = nullptr; const CPoint *C; size_t n;
    RowSum(const CPoint *C_, size_t n_);
    RowSum(const RowSum &) = delete; RowSum &operator=(const RowSum &) = delete;
    ~RowSum();
    Pt finish(const Fr *s);
};
// Group equations that do not feed the transcript — most of the verifier's arithmetic: per sum-check round two checks with a
// variable-base scalar multiplication each — are handed to a few background threads AS THEY ARISE, while the calling thread walks on
// through the rounds (whose hashed commitments are the sequential path); finish() drains what is left and reports the first failure.
// Lock-free: the producer fills a slot and bumps `count`; workers claim slots with a compare-exchange on `next`.
class Deferred {
public:
    Deferred() {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned nw = hw >= 12 ? 6 : hw >= 6 ? hw - 3 : 0;                // beside the caller and its spinning helpers (pool.h)
        if (const char *e = getenv("OTTI_VERIFY_THREADS")) { int v = atoi(e); if (v >= 0 && v <= 32) nw = (unsigned)v; }
        try { for (unsigned i = 0; i < nw; i++) th_.emplace_back([this] { work(false); }); } catch (const std::system_error &) {}
    }
    Deferred(const Deferred &) = delete; Deferred &operator=(const Deferred &) = delete;
    ~Deferred() { closing_.store(true, std::memory_order_release); for (auto &t : th_) t.join(); }
    void push_back(std::function<void()> f) {
        const size_t c = count_.load(std::memory_order_relaxed);
        if (c >= kSlots) { run_one(f); return; }                          // (never in practice: two checks per round, at most 2 x 64 rounds + a few)
        items_[c] = std::move(f);
        count_.store(c + 1, std::memory_order_release);
    }
    void finish() {                                                       // throws VerifyFail
        closing_.store(true, std::memory_order_release);
        work(true);
        for (auto &t : th_) t.join();
        th_.clear();
        if (const int c = code_.load()) throw VerifyFail{c};
    }
private:
    static constexpr size_t kSlots = 512;
    std::function<void()> items_[kSlots];
    std::atomic<size_t> count_{0}, next_{0};
    std::atomic<bool> closing_{false};
    std::atomic<int> code_{0};
    std::vector<std::thread> th_;
    void run_one(const std::function<void()> &f) {
        try { f(); } catch (const VerifyFail &e) { int z = 0; code_.compare_exchange_strong(z, e.code); } catch (...) { int z = 0; code_.compare_exchange_strong(z, (int)OTTI_ERR_VERIFY_INTERNAL); }
    }
    void work(bool until_empty) {
        for (;;) {
            size_t i = next_.load(std::memory_order_relaxed);
            if (i < count_.load(std::memory_order_acquire)) { if (next_.compare_exchange_weak(i, i + 1, std::memory_order_acq_rel)) run_one(items_[i]); continue; }
            if (until_empty || closing_.load(std::memory_order_acquire)) { if (next_.load() >= count_.load(std::memory_order_acquire)) return; continue; }
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
    }
};
void dotproductlog_verify(const DotProductProofLog &pf, size_t n, const Gens &g, const PcView &v, Transcript &tr, const Fr *a, const CPoint &Cx, const CPoint &Cy, Deferred *later = nullptr);

// GPU (snark_prover.cpp)
struct SnarkTimings { double ms[10]; };                      // the six R1CSProof stages, [6] derefs commitment, [7] product circuits, [8] hash layer, [9] total
std::unique_ptr<CompComm> snark_encode_gpu(Instance &inst, SnarkGens &g);
std::vector<uint8_t> snark_prove_gpu(Instance &inst, CompComm &comm, const uint8_t *vars32, size_t nvars, const std::vector<Fr> &inputs, SnarkGens &g,
                                     const void *tlabel, size_t tlabel_len, const uint8_t *seed32, SnarkTimings *tm);
struct DeviceWitness;                                        // device.h: the assignment resident in HBM (otti_witness_upload)
class ShardComm;                                             // shard.h
// sh != nullptr: this rank's part of ONE proof over the GPUs of a node (every rank passes the same inputs and gets the same bytes): the
// R1CS satisfiability proof sharded as in NIZK mode, the rows of the derefs commitment split over the ranks, the product circuits (hash layer,
// product layers, the device rounds of the batched sum-checks) by residue classes of the element index; the host rounds and the evaluation
// proofs run on every rank alike
std::vector<uint8_t> snark_prove_resident(Instance &inst, CompComm &comm, DeviceWitness &wit, SnarkGens &g, const void *tlabel, size_t tlabel_len,
                                          const uint8_t *seed32, SnarkTimings *tm, ShardComm *sh = nullptr);

}  // namespace otti
