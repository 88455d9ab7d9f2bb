// SNARK mode on the MI355X: SNARK::encode (computation commitment) and SNARK::prove = R1CSProof (prover.cpp, the NIZK device path) +
// R1CSEvalProof.  Follows upstream libspartan src/lib.rs, src/r1csinstance.rs, src/sparse_mlpoly.rs, src/product_tree.rs step for step
// [RECALL; /root/reference/Spartan is an empty submodule]; see snark.h.
//
// Data flow of R1CSEvalProof: the dense representation of A, B, C (addresses, timestamps, values: 16 N + 2 M field elements) and the
// two generator window tables stay in HBM; per proof the device builds eq(rx), eq(ry), dereferences them by address (6 N elements, committed
// with the bulk MSM), hashes 12 operation vectors and 4 memory vectors into product circuits (about 24 N + 8 M elements with all layers),
// and plays the layered sum-checks: every round is ONE launch over all tables of the batch (fold by the previous challenge + this round's
// sums, pcbatch_prove below), the last rounds of every layer on the host; the host only hashes four scalars per round (this sum-check is
// not zero-knowledge: no commitments on the sequential path).  The three closing
// polynomial-evaluation proofs reuse the log-size dot-product prover of NIZK mode (bullet rounds on the original generators).
#include "snark.h"
#include <atomic>
#include <sched.h>
#include "snark_dev.h"
#include "shard.h"
#include "pool.h"
#include "hosttail.h"
#include <chrono>
#include <functional>
#include <thread>
#include <condition_variable>
#include <mutex>

namespace otti {

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int kSumSlot = 64;                                  // where a round's sums land in the pinned result buffer

// the dense representation of sparse_mlpoly.rs MultiSparseMatPolynomialAsDense, resident in HBM
struct DeviceDecomm {
    size_t N = 0, M = 0;
    DevBuf<uint32_t> row_addr[3], col_addr[3];
    DevBuf<Fr> comb_ops;                                     // [row addr A,B,C | row read_ts A,B,C | col addr A,B,C | col read_ts A,B,C | val A,B,C | 0..]: 16 N
    DevBuf<Fr> comb_mem;                                     // [row audit_ts | col audit_ts]: 2 M
    Fr *part(int p, int k) const { return comb_ops.p + (size_t)(3 * p + k) * N; }
};

// Per-proof buffers (2.5 GB at 2^20: dereferenced values, sixteen product circuits with all their layers, dot-product tables, ...) are
// kept with the calling thread's device context and only ever grow: allocating and freeing them per proof cost ~4 ms of a 34 ms proof.
struct SnarkScratch {
    std::vector<DevBuf<Fr>> b;
    Fr *get(size_t slot, size_t n) { if (b.size() <= slot) b.resize(slot + 1); if (b[slot].n < n) b[slot].alloc(n); return b[slot].p; }
};
enum { SS_MEM_RX = 0, SS_MEM_RY, SS_EQS, SS_DEREFS, SS_PARTIALS, SS_DOTP, SS_PYR, SS_EO, SS_EM, SS_PE0 /* 11 slots */, SS_OPS = SS_PE0 + 11 /* 12 */, SS_MEMC = SS_OPS + 12 /* 4 */, SS_AHEAD = SS_MEMC + 4 /* 6: chunk sums of two bounds, their left and right eq tables */ };
static SnarkScratch &snark_workspace(DevCtx &c) { if (!c.snark_scratch) c.snark_scratch = new SnarkScratch(); return *c.snark_scratch; }

namespace {
// sparse_hint: 1 / 0 = the scalars are / are not mostly small numbers (picks the bulk MSM variant); -1 = look (a pass over Z and a synchronise)
std::vector<CPoint> commit_poly(DevCtx &c, Gens &gens, const Fr *Z, const PcSet &s, int sparse_hint = -1) {
    ensure_gens_device(gens);
    const bool sparse = sparse_hint >= 0 ? sparse_hint != 0 : dev_small_fraction(c, Z, s.L * s.R) > 0.25;
    dev_msm_rows(c, *gens.dev, Z, s.R, s.R, s.L, nullptr, nullptr, 0, MSM_COMPRESSED, nullptr, sparse);
    c.sync();
    std::vector<CPoint> out(s.L);
    memcpy(out.data(), c.h_points, 32 * s.L);
    return out;
}
void append_poly_commitment(Transcript &tr, const char *label, const std::vector<CPoint> &C) {
    tr.append_message(label, "poly_commitment_begin", 21);
    for (auto &p : C) tr.append_point("poly_commitment_share", p.b);
    tr.append_message(label, "poly_commitment_end", 19);
}
void append_unipoly(Transcript &tr, const Fr *c, size_t n) {
    tr.append_message("poly", "UniPoly_begin", 13);
    for (size_t i = 0; i < n; i++) tr.append_scalar("coeff", c[i]);
    tr.append_message("poly", "UniPoly_end", 11);
}
Fr reduce_evals(std::vector<Fr> v, const std::vector<Fr> &ch) {    // bound_poly_var_bot, last challenge first
    for (size_t i = ch.size(); i-- > 0;) { size_t h = v.size() / 2; for (size_t k = 0; k < h; k++) v[k] = fr_add(v[2 * k], fr_mul(ch[i], fr_sub(v[2 * k + 1], v[2 * k]))); v.resize(h); }
    return v[0];
}
}  // namespace

// ================================================================================================ SNARK::encode
std::unique_ptr<CompComm> snark_encode_gpu(Instance &I, SnarkGens &g) {
    DevCtx &c = DevCtx::get();
    if (I.num_cons != g.num_cons || I.num_vars != g.num_vars) throw Error(OTTI_ERR_BAD_ARG, "SNARK generators were made for a different instance size");
    size_t nz = 0; for (int k = 0; k < 3; k++) nz = std::max(nz, I.M[k].val.size() + ((I.given_cons < 2 && I.num_cons > I.M[k].val.size()) ? I.num_cons - I.M[k].val.size() : 0));
    const size_t N = next_pow2(std::max<size_t>(nz, 2)), M = (size_t)1 << std::max(ilog2(I.num_cons), ilog2(2 * I.num_vars));
    if (ilog2(16 * N) != g.ops.num_vars || ilog2(2 * M) != g.mem.num_vars) throw Error(OTTI_ERR_BAD_ARG, "SNARK generators were made for a different number of non-zero entries");
    auto dec = std::make_shared<DeviceDecomm>(); dec->N = N; dec->M = M;
    const bool trace = getenv("OTTI_TRACE") != nullptr; double t_lap = now_ms();
    auto lap = [&](const char *what) { if (!trace) return; c.sync(); const double t = now_ms(); fprintf(stderr, "[otti] snark_encode %-34s %.3f ms\n", what, t - t_lap); t_lap = t; };
    // MultiSparseMatPolynomialAsDense.  The host does the part that is a sequential scan over small integers (read_ts = visits of that
    // address so far, audit_ts = visits in total, shared by A, B, C) on u32 lists; their expansion to field elements (16 N + 2 M Montgomery
    // products) happens on the device, next to where the lists are needed anyway as gather indices.
    std::vector<uint32_t> addr[2][3], read_ts[2][3], audit[2];
    for (int k = 0; k < 3; k++) {
        const SparseMat &m = I.M[k];
        addr[0][k].assign(N, 0); addr[1][k].assign(N, 0);
        for (size_t i = 0; i < m.val.size(); i++) { addr[0][k][i] = m.row[i]; addr[1][k][i] = m.col[i]; }
        if (I.given_cons < 2)                                 // upstream pads 0 / 1 constraints with explicit zero entries (row i, column num_vars)
            for (size_t i = m.val.size(); i < I.num_cons; i++) { addr[0][k][i] = (uint32_t)i; addr[1][k][i] = (uint32_t)I.num_vars; }
    }
    auto scan_side = [&](int side) {                          // AddrTimestamps::new: audit_ts is shared by the three matrices
        audit[side].assign(M, 0);
        for (int k = 0; k < 3; k++) {
            read_ts[side][k].resize(N);
            for (size_t i = 0; i < N; i++) read_ts[side][k][i] = audit[side][addr[side][k][i]]++;
        }
    };
    { std::thread other(scan_side, 1); scan_side(0); other.join(); }
    lap("address / timestamp scans (host)");
    dec->comb_ops.alloc(16 * N); dec->comb_mem.alloc(2 * M);
    {
        DevBuf<uint32_t> tmp(std::max(N, M));
        OTTI_HIP(hipMemsetAsync(dec->comb_ops.p + 12 * N, 0, 4 * N * sizeof(Fr), c.stream));      // values (zero-padded) and the unused sixteenth part
        for (int k = 0; k < 3; k++) {
            dec->row_addr[k].alloc(N); dec->col_addr[k].alloc(N);
            OTTI_HIP(hipMemcpyAsync(dec->row_addr[k].p, addr[0][k].data(), N * 4, hipMemcpyHostToDevice, c.stream));
            OTTI_HIP(hipMemcpyAsync(dec->col_addr[k].p, addr[1][k].data(), N * 4, hipMemcpyHostToDevice, c.stream));
            dev_u32_to_fr(c, dec->row_addr[k].p, dec->part(0, k), N); dev_u32_to_fr(c, dec->col_addr[k].p, dec->part(2, k), N);
            for (int side = 0; side < 2; side++) {
                OTTI_HIP(hipMemcpyAsync(tmp.p, read_ts[side][k].data(), N * 4, hipMemcpyHostToDevice, c.stream));
                dev_u32_to_fr(c, tmp.p, dec->part(2 * side + 1, k), N);
                OTTI_HIP(hipStreamSynchronize(c.stream));     // tmp is reused (the copies come from pageable memory: nothing to overlap with)
            }
            if (!I.M[k].val.empty()) OTTI_HIP(hipMemcpyAsync(dec->part(4, k), I.M[k].val.data(), I.M[k].val.size() * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
        }
        for (int side = 0; side < 2; side++) {
            OTTI_HIP(hipMemcpyAsync(tmp.p, audit[side].data(), M * 4, hipMemcpyHostToDevice, c.stream));
            dev_u32_to_fr(c, tmp.p, dec->comb_mem.p + (size_t)side * M, M);
            OTTI_HIP(hipStreamSynchronize(c.stream));
        }
    }
    lap("uploads + expansion to field elements");
    auto cc = std::make_unique<CompComm>();
    cc->num_cons = I.num_cons; cc->num_vars = I.num_vars; cc->num_inputs = I.num_inputs; cc->num_ops = N; cc->num_mem_cells = M;
    cc->comm_ops = commit_poly(c, *g.eval, dec->comb_ops.p, g.ops);          // SparseMatPolynomial::multi_commit: comb_ops.commit(gens_ops, None), comb_mem.commit(gens_mem, None)
    lap("commit comb_ops (+ window table)");
    cc->comm_mem = commit_poly(c, *g.eval, dec->comb_mem.p, g.mem);
    lap("commit comb_mem");
    cc->dec = dec;
    return cc;
}

// ================================================================================================ product circuits and their batched proof
namespace {
// set when a persistent tail never answered (TailTimeout): the process goes on with a launch per round
std::atomic<bool> g_tail_off{false};
thread_local bool t_sharded_proof = false;                  // this thread is inside a sharded SNARK::prove (no persistent tail: see snark_prove_resident)
bool shard_comm_active() { return t_sharded_proof; }
// a batch of product circuits of one size: layer k of circuit i is (left, right) = store + off[k] + {0, n >> (k + 1)}.
// Sharded (G ranks, rank rk): the device holds this rank's residue class of every layer whose sides have at least G elements (element i' here
// is element i' G + rk there: every layer pairs i with i + side / 2, a multiple of G) — n_dev = n / G elements per circuit input, nl_dev layers;
// the few layers above them (sides shorter than G) exist on the host only.  small[k][i] = (left, right) of layer k in full, for every layer
// the host plays by itself (sides of at most kSmallSide elements), gathered from the ranks or computed from the layer below.
constexpr size_t kSmallSide = 256;                          // >= the longest host tail (2^OTTI_PC_LGT_*: 64 / 128 by default, hosttail.h takes up to 256)
struct Circuits {
    size_t n = 0, nl = 0; int count = 0;
    int G = 1, rk = 0; size_t n_dev = 0, nl_dev = 0;
    std::vector<Fr *> store; std::vector<size_t> off;
    std::vector<std::vector<std::pair<std::vector<Fr>, std::vector<Fr>>>> small;   // [layer][circuit], filled by gather_small() when sharded
    void init(int cnt, size_t n_, SnarkScratch &W, size_t first_slot, int G_ = 1, int rk_ = 0) {
        n = n_; count = cnt; nl = std::max<size_t>(1, ilog2(n)); G = G_; rk = rk_; n_dev = n / (size_t)G; nl_dev = std::max<size_t>(1, ilog2(n_dev));
        store.resize(cnt); off.assign(nl_dev, 0);
        size_t o = 0; for (size_t k = 0; k < nl_dev; k++) { off[k] = o; o += n_dev >> k; }
        for (int i = 0; i < cnt; i++) store[i] = W.get(first_slot + i, o);
    }
    size_t side(size_t k) const { return n >> (k + 1); }                 // GLOBAL elements per side of layer k
    size_t side_dev(size_t k) const { return n_dev >> (k + 1); }         // ... of which this rank holds
    Fr *left(int i, size_t k) { return store[i] + off[k]; }
    Fr *right(int i, size_t k) { return store[i] + off[k] + (n_dev >> (k + 1)); }
    Fr *input(int i) { return store[i]; }                     // layer 0: the hashed vector itself, left half then right half
    void build(DevCtx &c) {                                   // ProductCircuit::new: compute_layer, all circuits of the batch per launch
        for (size_t k = 1; k < nl_dev; k++) {
            LayerList L; L.n = count;
            for (int i = 0; i < count; i++) { L.in_left[i] = left(i, k - 1); L.in_right[i] = right(i, k - 1); L.out_left[i] = left(i, k); L.out_right[i] = right(i, k); }
            dev_prod_layer(c, L, n_dev >> (k + 1));
        }
    }
    // sharded: the layers the host plays alone, in full on every rank
    void gather_small(DevCtx &c, ShardComm &sh) {
        small.assign(nl, {});
        for (size_t k = 0; k < nl; k++) {
            const size_t h = side(k);
            if (h > kSmallSide) continue;
            small[k].resize(count);
            if (k < nl_dev) {                                 // on the devices: every rank's share, interleaved
                const size_t hl = side_dev(k);
                std::vector<Fr> mine(2 * hl * count), all(mine.size() * (size_t)G);
                for (int i = 0; i < count; i++) {
                    OTTI_HIP(hipMemcpyAsync(&mine[(size_t)2 * i * hl], left(i, k), hl * sizeof(Fr), hipMemcpyDeviceToHost, c.stream));
                    OTTI_HIP(hipMemcpyAsync(&mine[(size_t)(2 * i + 1) * hl], right(i, k), hl * sizeof(Fr), hipMemcpyDeviceToHost, c.stream));
                }
                OTTI_HIP(hipStreamSynchronize(c.stream));
                sh.allgather(mine.data(), mine.size() * sizeof(Fr), all.data());
                for (int i = 0; i < count; i++) {
                    auto &lr = small[k][i]; lr.first.resize(h); lr.second.resize(h);
                    for (int r = 0; r < G; r++) for (size_t e = 0; e < hl; e++) {
                        lr.first[e * G + r] = all[(size_t)r * mine.size() + (size_t)2 * i * hl + e];
                        lr.second[e * G + r] = all[(size_t)r * mine.size() + (size_t)(2 * i + 1) * hl + e];
                    }
                }
            } else {                                          // above the devices' layers: compute_layer on the host from the layer below
                for (int i = 0; i < count; i++) {
                    const auto &below = small[k - 1][i]; auto &lr = small[k][i]; lr.first.resize(h); lr.second.resize(h);
                    for (size_t e = 0; e < h; e++) { lr.first[e] = fr_mul(below.first[e], below.second[e]); lr.second[e] = fr_mul(below.first[h + e], below.second[h + e]); }
                }
            }
        }
    }
};
struct DotpTables { Fr *l[6], *r[6], *w[6]; size_t len = 0; int n = 0; };

// ProductCircuitEvalProofBatched::prove.  evals: the circuits' outputs (already known to the caller).  The tables are folded in place.
//
// One layer = one SumcheckInstanceProof::prove_cubic_batched over (left, right, eq(rand)) of every circuit (+ the dot-product triples at
// the input layer).  Per round ONE launch (k_pc_round: fold by the previous challenge + this round's sums, mailed to the host by the
// last workgroup).  The eq table is never stored: eq(rand, .) is a tensor product, so after j rounds it is
//   c_j * (b ? rand_j : 1 - rand_j) * E_j[i],  E_j = eq(rand[j+1..), .),  c_j = prod_{k<j} eq(rand_k, r_k)
// (two L2-resident pyramids of small tables, as phase one of the R1CS proof); the kernel returns S_t = sum_i E_j[i] (A_t B_t)[i] and the
// host applies c_j * ((1 - rand_j) + t (2 rand_j - 1)).  Once the tables are down to T elements they are exported to pinned memory and
// the host plays the last rounds itself: a launch + hand-off costs more than the arithmetic of such a round on a host core.
constexpr int kPcTailSlot = 128;
constexpr int kPcPreExportEnd = 7400;                       // pre-exported host-only layers end below the hash layer's ahead-of-time results (kHashEvalSlot)
// sh (sharded SNARK::prove): the tables are this rank's residue classes (Circuits above; D: strided copies); a device round works on them
// with the eq factor taken at the global index, its sums are added across the ranks (allreduce_fr: 3 elements per instance), and where the
// host takes a layer over the ranks' shares of its tables are gathered and interleaved.  Host rounds run on every rank alike.
ProductCircuitEvalProofBatched pcbatch_prove(DevCtx &c, Circuits &C, const std::vector<Fr> &evals, DotpTables *D, const std::vector<Fr> &dotp_evals, Transcript &tr,
                                             Fr *pyr, std::vector<Fr> &rand_out, ShardComm *sh = nullptr, const std::function<void()> *on_start = nullptr) {
    const int np = C.count; const size_t nl = C.nl;
    const size_t G = sh ? (size_t)sh->world() : 1, rk = sh ? (size_t)sh->rank() : 0;
    if (sh && (C.G != (int)G || C.small.size() != nl)) throw Error(OTTI_ERR_INTERNAL, "sharded product circuits were not prepared for this exchange");
    ProductCircuitEvalProofBatched pf; pf.layers.resize(nl);
    std::vector<Fr> claims = evals, rand, rprod;
    const Fr one = fr_one();
    struct Release { DevCtx &c; ~Release() { c.go_abort(); } } release{c};      // an exception below must not leave an armed kernel waiting
    const bool arm_ok = c.armed_ok();
    // OTTI_PC_TAIL=0 switches the persistent tail off; OTTI_PC_TAIL_CAP shrinks its per-workgroup capacity (tests: small instances then
    // also take the route where the tail picks up tables that earlier launches folded in HBM)
    static const bool tail_env = [] { const char *e = getenv("OTTI_PC_TAIL"); return !(e && e[0] == '0'); }();
    static const size_t tail_cap = [] { const char *e = getenv("OTTI_PC_TAIL_CAP"); size_t v = e ? (size_t)atoi(e) : 0; return (v >= 2 && v <= (size_t)kTailCap && !(v & (v - 1))) ? v : (size_t)kTailCap; }();
    // elements of a table a workgroup of the tail starts with (it spreads wider only for what does not fit): fewer, busier workgroups mean fewer
    // mail lines per round for the host to collect — the larger cost (tools/hosttail_variants.sh)
    static const size_t tail_per_wg = [] { const char *e = getenv("OTTI_PC_TAIL_PER_WG"); size_t v = e ? (size_t)atoi(e) : 0; return (v >= 16 && v <= (size_t)kTailCap && !(v & (v - 1))) ? v : (size_t)128; }();
    const bool tail_ok = arm_ok && tail_env && !g_tail_off.load(std::memory_order_relaxed) && !shard_comm_active();
    // its grid (W workgroups per instance, one per CU: 96 KB of LDS each) must be resident as a whole: never more workgroups than the device has CUs
    const int tail_groups_max = std::min(kTailMaxGroups, c.num_cu);
    static const size_t lgt_env_many = [] { const char *e = getenv("OTTI_PC_LGT_MANY"); return e ? (size_t)atoi(e) : (size_t)0; }();
    static const size_t lgt_env_few = [] { const char *e = getenv("OTTI_PC_LGT_FEW"); return e ? (size_t)atoi(e) : (size_t)0; }();
    static const size_t pc_arm_max = [] { const char *e = getenv("OTTI_PC_ARM_MAX"); return e ? (size_t)atoll(e) : (size_t)1 << 22; }();     // (the sum-check kernels of the R1CS proof arm up to kArmMaxLen; here a round more or less ahead costs nothing else)
    SpinPool &pool = SpinPool::get();
    const int host_threads = std::min(8, pool.workers() + 1);
    // How many of a layer's last rounds the host plays (hosttail.h).  With the AVX-512 IFMA form a host round over tables of 32 / 64 elements costs
    // less than the 16 us of a round of the persistent launch: the last 6 rounds of the 12- and 18-instance batches (tables of 64), the last 7 of
    // the 4-instance batches (128); measured 4/5, 5/6, 5/7, 5/8, 6/7: product circuits 9.5, 9.1, 9.15, 9.3, 9.0 ms (tools/hosttail_variants.sh,
    // profiles/r4_hosttail_variants.txt).  With the scalar form (no such instructions): 4 and 5 as before (5/7 cost 10.9-11.4 ms against 10.7).
    const bool fr8 = host_fr8_available();
    const size_t lgt_many = lgt_env_many ? lgt_env_many : (fr8 ? 6 : 4), lgt_few = lgt_env_few ? lgt_env_few : (fr8 ? 7 : 5);
    static const bool trace = getenv("OTTI_TRACE") != nullptr;
    double tr_tail_first_ms = 0, tr_tail_sum_ms = 0, tr_tail_wait_ms = 0, tr_tail_ms = 0, tr_launch_ms = 0, tr_host_ms = 0, tr_layer0_ms = 0; size_t tr_tail_rounds = 0, tr_launch_rounds = 0, tr_host_rounds = 0, tr_tail_layers = 0;
    // what a layer's rounds are made of, decided from its position alone (so that the NEXT layer's first launches can be issued ahead of time)
    struct Plan { size_t layer_id = 0, nr = 0, h = 1; bool with_dotp = false, on_device = true; PcList P; int ni = 0; size_t lgT = 0, T = 1, ndev = 0, k0 = 0; int tailW = 1; bool tail = false; };
    auto make_plan = [&](size_t li_, size_t nr_) {
        Plan p; p.layer_id = nl - 1 - li_; p.nr = nr_; p.h = (size_t)1 << nr_;      // elements per side in this layer, one round per variable
        p.with_dotp = p.layer_id == 0 && D && D->n;
        p.on_device = p.layer_id < C.nl_dev;                     // (sharded: the layers with sides shorter than the number of ranks exist on the host only)
        p.P.n = 0;
        for (int i = 0; i < np; i++) { p.P.A[p.P.n] = p.on_device ? C.left(i, p.layer_id) : nullptr; p.P.B[p.P.n] = p.on_device ? C.right(i, p.layer_id) : nullptr; p.P.C[p.P.n] = nullptr; p.P.n++; }
        if (p.with_dotp) for (int i = 0; i < D->n; i++) { p.P.A[p.P.n] = D->l[i]; p.P.B[p.P.n] = D->r[i]; p.P.C[p.P.n] = D->w[i]; p.P.n++; }
        p.ni = p.P.n;
        p.lgT = std::min<size_t>(nr_, p.ni >= 8 ? lgt_many : lgt_few); p.T = (size_t)1 << p.lgT; p.ndev = nr_ - p.lgT;
        // The persistent tail (k_pc_tail, snark_dev.h): from round k0 on — the first round whose tables fit the LDS of W workgroups per
        // instance — ONE launch plays every remaining device round.  Only while this is the process's single proof in flight (its grid
        // must be resident as a whole: the workgroups wait for the host, the host for all of them) and no kernel class it belongs to is
        // being timed; otherwise, and for the rounds before k0, a launch per round as before.
        p.k0 = p.ndev; p.tailW = 1;
        if (p.ndev && tail_ok && p.ni <= tail_groups_max) {
            int Wmax = 1; while (2 * Wmax * p.ni <= tail_groups_max && (size_t)(2 * Wmax) <= p.T) Wmax *= 2;
            const size_t cap_all = tail_cap * (size_t)Wmax;
            p.k0 = 0; while ((p.h >> p.k0) > cap_all) p.k0++;
            if (p.k0 >= p.ndev) p.k0 = p.ndev;                   // (cannot happen for cap_all >= 2 T; kept for a shrunken test capacity)
            else { const size_t len0 = p.h >> p.k0; p.tailW = 1; while (p.tailW < Wmax && len0 / (size_t)p.tailW > tail_per_wg) p.tailW *= 2; while (len0 / (size_t)p.tailW > tail_cap) p.tailW *= 2; }
        }
        p.tail = p.k0 < p.ndev;
        return p;
    };
    // eq(rand[1..], .) of a layer with nr variables as the round kernels read it: pyramids over the last n_lo of rand[1..] and the n_hi before them
    Fr *const pyr_lo = pyr, *const pyr_hi = pyr + 8192;
    auto eq_src_of = [&](size_t nr_, size_t m, const Fr *first_var) {
        const size_t nv = nr_ ? nr_ - 1 : 0, n_lo = std::min<size_t>(nv, 12);
        EqSrc e;
        const size_t mt = std::min(m, nv);                       // tabulated variables
        if (mt <= n_lo) { e.hi = nullptr; e.lo = pyr_lo + (((size_t)1 << mt) - 1); e.lo_bits = 0; }
        else { e.hi = pyr_hi + (((size_t)1 << (mt - n_lo)) - 1); e.lo = pyr_lo + (((size_t)1 << n_lo) - 1); e.lo_bits = (int)n_lo; }
        if (m > nv) { if (m != nv + 1 || !nr_ || !first_var) throw Error(OTTI_ERR_INTERNAL, "eq table over more variables than the layer has"); e.top_bit = (int)nv; e.top = *first_var; }
        e.stride = (uint32_t)G; e.offset = (uint32_t)rk;           // sharded: item i of a kernel is element i G + rk of the table
        return e;
    };
    auto launch_pyramids = [&](size_t nr_, const Fr *vars_from_1 /* nr_ - 1 of them */) {
        const size_t nv = nr_ ? nr_ - 1 : 0, n_lo = std::min<size_t>(nv, 12), n_hi = nv - n_lo;
        if (n_hi > 13) throw Error(OTTI_ERR_BAD_ARG, "product circuit over more than 2^26 elements");
        dev_eq_pyramid2(c, vars_from_1 + n_hi, n_lo, pyr_lo, vars_from_1, n_hi, n_hi ? pyr_hi : nullptr);
    };
    // launched at the end of the layer before (see there): the pyramids of the layer about to start, and its first round's kernel
    bool pyr_ahead = false, eval_ahead = false; unsigned long long eval_ahead_tick = 0;
    // The layers the host plays alone (sides of at most T elements: the first 7 / 8 layers) are exported to pinned memory by ONE run of tiny launches
    // queued here, each to a place of its own — not one launch and one wait at the head of each layer (layer li has li variables whatever the
    // challenges are).  on_start (the caller's work for a second stream) is queued after them, so that they are not held up behind it.
    // OTTI_PC_PREEXPORT=0: a launch per layer as before.
    static const bool preexport_env = [] { const char *e = getenv("OTTI_PC_PREEXPORT"); return !(e && e[0] == '0'); }();
    std::vector<unsigned long long> pre_tick(nl, 0); std::vector<int> pre_slot(nl, kPcTailSlot);
    if (!sh && preexport_env) {
        size_t at = kPcTailSlot;
        for (size_t li = 0; li < nl; li++) {
            const Plan p = make_plan(li, li);
            const size_t need = (size_t)3 * p.ni * p.h;
            if (p.ndev || !p.on_device || at + need > (size_t)kPcPreExportEnd) break;
            pre_slot[li] = (int)at; pre_tick[li] = dev_pc_export(c, p.P, p.h, false, nullptr, (int)at); at += need;
        }
    }
    if (on_start) (*on_start)();
    for (size_t li = 0; li < nl; li++) {
        const double tr_layer_start = trace ? now_ms() : 0;
        Plan plan = make_plan(li, rand.size());
        const size_t layer_id = plan.layer_id, nr = plan.nr, h = plan.h, lgT = plan.lgT, T = plan.T, ndev = plan.ndev, k0 = plan.k0;
        const bool with_dotp = plan.with_dotp, on_device = plan.on_device, tail = plan.tail;
        const PcList &P = plan.P; const int ni = plan.ni, tailW = plan.tailW;
        (void)lgT;
        if (with_dotp) { if (D->len != h / G) throw Error(OTTI_ERR_INTERNAL, "dot-product circuits do not match the input layer"); claims.insert(claims.end(), dotp_evals.begin(), dotp_evals.end()); }
        if (sh && ndev && (T < G || !on_device)) throw Error(OTTI_ERR_INTERNAL, "sharded product circuits: more ranks than a host tail has elements");
        unsigned long long tail_seq = 0;
        // The layer's FIRST variable is in no pyramid: no round's factor table contains it (round j uses eq over rand[j+1..]), only the persistent
        // tail's own eq table when it starts at round 0 (EqSrc.top) — and rand[0] is the last challenge to be drawn, so without it the pyramids of
        // layer li + 1 (and its first round's kernel, when that is a launch of its own) are issued as soon as layer li's last round challenge is
        // out (below) and run while the host absorbs the layer's claims and draws the next coefficients.
        if (ndev && !pyr_ahead) launch_pyramids(nr, rand.data() + 1);
        pyr_ahead = false;
        auto eq_src = [&](size_t m) { return eq_src_of(nr, m, nr ? rand.data() : nullptr); };
        // launch k >= 1 folds by r_{k-1} and yields the sums of round k (k < ndev) or the exported tail (k == ndev).  Armed (device.h), it is
        // queued one round ahead and starts the moment the host publishes r_{k-1}.
        auto armed = [&](size_t k) { return arm_ok && k >= 1 && k < k0 + (tail ? 0 : 1) && k <= ndev && (h >> (k - 1)) * (size_t)ni <= pc_arm_max; };   // small grids only (device.h); never the tail's own launch
        std::vector<unsigned long long> tick(ndev + 2, 0);
        auto launch_tail = [&](const Fr *r) { tail_seq = dev_pc_tail(c, P, tailW, h >> k0, T, r, eq_src(nr - k0), kPcTailSlot); };
        auto launch_for = [&](size_t k, const Fr *r) {
            const size_t len_in = (h >> (k - 1)) / G;         // of this rank
            if (tail && k == k0) { launch_tail(r); return; }
            tick[k] = k < ndev ? dev_pc_fold_eval(c, P, len_in, r, eq_src(nr - k - 1), kSumSlot) : dev_pc_export(c, P, len_in, true, r, kPcTailSlot);
        };
        if (tail && k0 == 0) launch_tail(nullptr);
        else if (ndev) { tick[0] = eval_ahead ? eval_ahead_tick : dev_pc_eval(c, P, h / G, eq_src(nr - 1), kSumSlot); if (armed(1)) launch_for(1, nullptr); }
        else if (!sh) tick[0] = pre_tick[li] ? pre_tick[li] : dev_pc_export(c, P, h, false, nullptr, kPcTailSlot);     // (sharded: the host-only layers are in C.small already)
        const int layer_slot = (!ndev && pre_tick[li]) ? pre_slot[li] : kPcTailSlot;      // where this layer's exported tables are
        eval_ahead = false;
        std::vector<Fr> coeff = tr.challenge_vector("rand_coeffs_next_layer", claims.size());
        Fr e = fr_zero(); for (size_t k = 0; k < claims.size(); k++) e = fr_add(e, fr_mul(claims[k], coeff[k]));
        LayerProofBatched &L = pf.layers[li];
        rprod.clear();
        std::vector<std::vector<Fr>> tA(ni), tB(ni), tC(ni); std::vector<Fr> tE; bool tail_built = false;   // host tail: T elements per table
        std::unique_ptr<HostTail> host_tail;
        Fr cj = one, cj_tail = one;                         // cj_tail: the eq factor accumulated before the tail took over (its eq table carries the rest)
        if (trace) { tr_layer0_ms += now_ms() - tr_layer_start; if (tail) tr_tail_layers++; }
        for (size_t j = 0; j < nr; j++) {                    // SumcheckInstanceProof::prove_cubic_batched
            const double tr_round_start = trace ? now_ms() : 0;
            Fr c0 = fr_zero(), c2 = fr_zero(), c3 = fr_zero();
            if (j < ndev && tail && j >= k0) {
                // a round of the persistent launch: W partial sums per instance, in the workgroups' own mail lines; the eq table is a real
                // third table there, so the sums already carry the bound variable's factor — only the factor of the rounds before k0 is missing
                if (j == k0) cj_tail = cj;
                Fr inst_sums[3 * kMaxInst];
                {
                    const double tw = trace ? now_ms() : 0;
                    if (trace) { c.wait_tail(1, tail_seq + (j - k0)); tr_tail_first_ms += now_ms() - tw; }     // (trace only: when the first line is in)
                    c.wait_tail_sums(ni, tailW, tail_seq + (j - k0), inst_sums);
                    if (trace) tr_tail_wait_ms += now_ms() - tw;
                }
                Fr ps[3], ds[3];                                      // the product instances' sums (they share the eq factor) and the triples', times the coefficients
                weighted_sums3(inst_sums, coeff.data(), np, ni, ps, ds);
                c0 = fr_add(ds[0], fr_mul(cj_tail, ps[0])); c2 = fr_add(ds[1], fr_mul(cj_tail, ps[1])); c3 = fr_add(ds[2], fr_mul(cj_tail, ps[2]));
                if (trace) tr_tail_sum_ms += now_ms() - tr_round_start;
            } else if (j < ndev) {
                c.wait_ticket(tick[j]);
                if (sh) sh->allreduce_fr(&c.h_results[kSumSlot], (size_t)3 * ni);      // this round's sums over the ranks' residue classes (pinned memory: the next launch writes them afresh)
                const Fr &tau = rand[j];
                const Fr w0 = fr_sub(one, tau), dw = fr_sub(fr_add(tau, tau), one), w2 = fr_add(w0, fr_add(dw, dw)), w3 = fr_add(w2, dw);
                const Fr f0 = fr_mul(cj, w0), f2 = fr_mul(cj, w2), f3 = fr_mul(cj, w3);
                Fr ps[3], ds[3];                                      // the product circuits share the eq factor
                weighted_sums3(&c.h_results[kSumSlot], coeff.data(), np, ni, ps, ds);
                c0 = fr_add(ds[0], fr_mul(f0, ps[0])); c2 = fr_add(ds[1], fr_mul(f2, ps[1])); c3 = fr_add(ds[2], fr_mul(f3, ps[2]));
            } else {
                if (!tail_built) {
                    bool direct = false;
                    if (sh && ndev == 0) {                          // a layer the host plays alone: in full on every rank already (product circuits only)
                        if (C.small[layer_id].size() != (size_t)ni || C.small[layer_id][0].first.size() != T) throw Error(OTTI_ERR_INTERNAL, "sharded product circuits: a host-played layer was not gathered (host tail longer than kSmallSide)");
                        for (int k = 0; k < ni; k++) { tA[k] = C.small[layer_id][k].first; tB[k] = C.small[layer_id][k].second; }
                    } else if (sh) {                                // every rank's share of the exported tables, interleaved: element e of rank r is element e G + r
                        c.wait_ticket(tick[ndev]);
                        const size_t Tl = T / G, per = (size_t)3 * ni * Tl;
                        std::vector<Fr> all(per * G);
                        sh->allgather(&c.h_results[kPcTailSlot], per * sizeof(Fr), all.data());
                        for (int k = 0; k < ni; k++) {
                            tA[k].resize(T); tB[k].resize(T); if (k >= np) tC[k].resize(T);
                            for (size_t r = 0; r < G; r++) for (size_t e = 0; e < Tl; e++) {
                                const Fr *base = &all[r * per + (size_t)3 * k * Tl];
                                tA[k][e * G + r] = base[e]; tB[k][e * G + r] = base[Tl + e];
                                if (k >= np) tC[k][e * G + r] = base[2 * Tl + e];
                            }
                        }
                    } else {
                    if (tail) c.wait_tail(ni * tailW, tail_seq + (ndev - k0)); else c.wait_ticket(tick[ndev]);
                    direct = true;                                  // the host tail packs the tables straight out of the pinned buffer the device exported them to
                    }
                    tE = eq_evals_host(rand.data() + ndev, nr - ndev);
                    for (auto &x : tE) x = fr_mul(x, cj);
                    std::vector<const Fr *> pa(ni), pb(ni), pc(ni);
                    for (int k = 0; k < ni; k++) {
                        const Fr *base = &c.h_results[layer_slot + (size_t)3 * k * T];
                        pa[k] = direct ? base : tA[k].data(); pb[k] = direct ? base + T : tB[k].data(); pc[k] = k < np ? nullptr : direct ? base + 2 * T : tC[k].data();
                    }
                    host_tail = HostTail::make(np, ni - np, T, pa.data(), pb.data(), pc.data(), tE.data(), coeff.data(), host_threads);   // hosttail.h: AVX-512 IFMA where the CPU has it
                    tail_built = true;
                }
                Fr hs[3]; host_tail->sums(hs);
                c0 = hs[0]; c2 = hs[1]; c3 = hs[2];
            }
            Fr evals4[4] = {c0, fr_sub(e, c0), c2, c3}, poly[4];
            unipoly_from_evals(poly, evals4, 4);
            append_unipoly(tr, poly, 4);
            const Fr r_j = tr.challenge_scalar("challenge_nextround");
            rprod.push_back(r_j);
            if (j + 1 == nr && li + 1 < nl && j >= ndev) {
                // the layer's last challenge: everything the NEXT layer's eq pyramids are made of (its variables 1 .. nr are this layer's challenges; its
                // variable 0 comes after the claims below and is in no pyramid).  The device is idle — this layer's last rounds are the host's — so
                // the pyramids, and the next layer's first sums when they are a launch of their own (they read tables and pyramids only), run under
                // the host's closing work and the next layer's coefficient draws instead of in front of its first round.
                const Plan nx = make_plan(li + 1, nr + 1);
                if (nx.ndev) {
                    launch_pyramids(nx.nr, rprod.data());
                    pyr_ahead = true;
                    if (!(nx.tail && nx.k0 == 0) && !(sh && (nx.T < G || !nx.on_device))) {
                        eval_ahead_tick = dev_pc_eval(c, nx.P, nx.h / G, eq_src_of(nx.nr, nx.nr - 1, nullptr), kSumSlot);
                        eval_ahead = true;
                    }
                }
            }
            if (j < ndev) {
                if (tail && j >= k0) c.go(&r_j, 1);              // the persistent launch folds and goes on (or exports, after its last round)
                else {
                    if (armed(j + 1)) c.go(&r_j, 1); else launch_for(j + 1, &r_j);
                    if (armed(j + 2)) launch_for(j + 2, nullptr);
                }
            }
            if (j < ndev) cj = fr_mul(cj, fr_add(fr_mul(rand[j], r_j), fr_mul(fr_sub(one, rand[j]), fr_sub(one, r_j))));
            else host_tail->fold(r_j);
            e = unipoly_eval(poly, 4, r_j);
            L.coeffs.push_back(poly[0]); L.coeffs.push_back(poly[2]); L.coeffs.push_back(poly[3]);      // UniPoly::compress
            if (trace) {
                const double dt = now_ms() - tr_round_start;
                if (j < ndev && tail && j >= k0) { tr_tail_ms += dt; tr_tail_rounds++; } else if (j < ndev) { tr_launch_ms += dt; tr_launch_rounds++; } else { tr_host_ms += dt; tr_host_rounds++; }
            }
        }
        if (host_tail) {                                    // the tables' last elements, out of the host tail's own representation
            if (host_tail->len() != 1) throw Error(OTTI_ERR_INTERNAL, "host sum-check tail ended early");
            for (int k = 0; k < ni; k++) { Fr t3[3]; host_tail->last(k, t3); tA[k].assign(1, t3[0]); tB[k].assign(1, t3[1]); if (k >= np) tC[k].assign(1, t3[2]); }
        }
        if (!tail_built && sh) {                             // a layer without rounds, sharded: from the host copies
            for (int k = 0; k < ni; k++) { tA[k].assign(1, C.small[layer_id][k].first[0]); tB[k].assign(1, C.small[layer_id][k].second[0]); }
        } else if (!tail_built) {                            // a layer without rounds: the tables are single elements
            c.wait_ticket(tick[0]);
            for (int k = 0; k < ni; k++) { const Fr *base = &c.h_results[layer_slot + (size_t)3 * k * T]; tA[k].assign(base, base + 1); tB[k].assign(base + T, base + T + 1); if (k >= np) tC[k].assign(base + 2 * T, base + 2 * T + 1); }
        }
        // the tables' last elements: claims_prod (left, right per circuit; the eq table's is not sent), then the dot-product triples
        L.left.resize(np); L.right.resize(np);
        for (int i = 0; i < np; i++) { L.left[i] = tA[i][0]; L.right[i] = tB[i][0]; tr.append_scalar("claim_prod_left", L.left[i]); tr.append_scalar("claim_prod_right", L.right[i]); }
        if (with_dotp) for (int i = 0; i < D->n; i++) {
            const Fr t[3] = {tA[np + i][0], tB[np + i][0], tC[np + i][0]};
            pf.dotp_left.push_back(t[0]); pf.dotp_right.push_back(t[1]); pf.dotp_weight.push_back(t[2]);
            tr.append_scalar("claim_dotp_left", t[0]); tr.append_scalar("claim_dotp_right", t[1]); tr.append_scalar("claim_dotp_weight", t[2]);
        }
        const Fr r_layer = tr.challenge_scalar("challenge_r_layer");
        claims.resize(np);
        for (int i = 0; i < np; i++) claims[i] = fr_add(L.left[i], fr_mul(r_layer, fr_sub(L.right[i], L.left[i])));
        std::vector<Fr> ext = {r_layer}; ext.insert(ext.end(), rprod.begin(), rprod.end()); rand = ext;
    }
    if (trace)
        fprintf(stderr, "[otti] pcbatch ni=%d layers=%zu (tail in %zu): %zu tail rounds %.3f ms (%.1f us each: first mail in after %.1f, all after %.1f, summed and combined after %.1f), %zu launch rounds %.3f ms (%.1f us each), %zu host rounds %.3f ms, layer set-up %.3f ms\n",
                np, nl, tr_tail_layers, tr_tail_rounds, tr_tail_ms, tr_tail_rounds ? 1e3 * tr_tail_ms / tr_tail_rounds : 0.0, tr_tail_rounds ? 1e3 * tr_tail_first_ms / tr_tail_rounds : 0.0, tr_tail_rounds ? 1e3 * tr_tail_wait_ms / tr_tail_rounds : 0.0, tr_tail_rounds ? 1e3 * tr_tail_sum_ms / tr_tail_rounds : 0.0, tr_launch_rounds, tr_launch_ms,
                tr_launch_rounds ? 1e3 * tr_launch_ms / tr_launch_rounds : 0.0, tr_host_rounds, tr_host_ms, tr_layer0_ms);
    rand_out = rand;
    return pf;
}

// PolyEvalProof::prove(poly, None, r, Zr, None, gens, ..) on a polynomial resident in HBM
// chunks (optional): the bound's chunk sums over eq of the left point's variables a .. (dev_poly_bound_chunks, computed ahead of time: the point's first
// a variables are the last to be drawn) — the bound is then a 2^a-row one over them
DotProductProofLog polyeval_prove_plain(DevCtx &c, Gens &gens, const PcSet &s, const Fr *Z, const std::vector<Fr> &r, const Fr &Zr, Transcript &tr, RandomTape &tape,
                                        const Fr *chunks = nullptr, size_t a_first = 0) {
    if (r.size() != s.num_vars) throw Error(OTTI_ERR_INTERNAL, "evaluation point of the wrong length");
    tr.append_protocol_name("polynomial evaluation proof");
    const size_t lv = s.num_vars / 2, lgR = ilog2(s.R);
    SnarkScratch &W = snark_workspace(c);
    Fr *Lv = W.get(SS_PE0 + 0, s.L), *Rv = W.get(SS_PE0 + 1, s.R), *LZ = W.get(SS_PE0 + 2, s.R), *a = W.get(SS_PE0 + 3, s.R), *sbuf = W.get(SS_PE0 + 4, s.R),
       *b2 = W.get(SS_PE0 + 5, s.R), *s2 = W.get(SS_PE0 + 6, s.R), *rows = W.get(SS_PE0 + 7, 2 * s.R), *extras = W.get(SS_PE0 + 8, 4 * (lgR + 1)),
       *eqs = W.get(SS_PE0 + 9, 5 * 4096), *bound = W.get(SS_PE0 + 10, 64 * s.R);
    if (chunks) {
        dev_eq_evals2(c, r.data(), a_first, Lv, r.data() + lv, s.num_vars - lv, Rv, eqs);
        dev_poly_bound(c, chunks, (size_t)1 << a_first, s.R, Lv, LZ, bound);
    } else {
        dev_eq_evals2(c, r.data(), lv, Lv, r.data() + lv, s.num_vars - lv, Rv, eqs);
        dev_poly_bound(c, Z, s.L, s.R, Lv, LZ, bound);
    }
    const PcView pv = {s.h_n, s.g1, s.h1, s.R};
    const PeBufs pb = {LZ, Rv, a, sbuf, b2, s2, rows, extras};
    CPoint Cy;
    return dplog_prove_device(c, *gens.dev, gens, pv, pb, fr_zero(), &Zr, fr_zero(), Cy, tr, tape);
}
}  // namespace

// ---- the row half of the derefs commitment, ahead of time.  The dereferenced polynomial is [eq(rx) at the row addresses of A, B, C |
// eq(ry) at the column addresses | zeros]: its first 3 N / R commitment rows depend on rx alone, which the R1CS proof fixes at the end
// of its FIRST sum-check — and everything the proof does after that (sigma protocols, second sum-check, evaluation proof: ~2 ms at 2^20)
// is a chain of latency-bound rounds that leaves the chip idle.  A helper thread with a device context of its own therefore sums those
// rows meanwhile (3.2 ms of the chip's multiplier), on a stream confined to 192 of the 256 CUs (hipExtStreamCreateWithCUMask) so that the rounds keep
// CUs to themselves — a fixed-base MSM workgroup holds its CU for ~2 ms, and without the mask a round's kernel waits for one to end;
// the column half is committed by the proving thread once ry is known, both launches sharing the chip.
struct RowsAhead {
    Gens &gens; const Fr *Z; size_t R, rows;
    std::thread th; std::mutex mu; std::condition_variable cv; int stage = 0;      // 1: the rows' scalars are on their way (ev recorded), -1: cancelled
    hipEvent_t ev = nullptr; std::vector<CPoint> C; std::exception_ptr err; bool done = false;
    RowsAhead(Gens &g, const Fr *Z_, size_t R_, size_t rows_) : gens(g), Z(Z_), R(R_), rows(rows_) {
        OTTI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        th = std::thread([this] { run(); });
    }
    ~RowsAhead() { { std::lock_guard<std::mutex> lk(mu); if (stage == 0) stage = -1; } cv.notify_all(); if (th.joinable()) th.join(); if (ev) (void)hipEventDestroy(ev); }
    bool recorded = false;
    void record(hipStream_t producer) { OTTI_HIP(hipEventRecord(ev, producer)); recorded = true; }   // the scalars are complete once `producer` reaches this point
    void release() {                                              // start summing (after record)
        if (!recorded) return;
        { std::lock_guard<std::mutex> lk(mu); if (stage == 0) stage = 1; }
        cv.notify_all();
    }
    std::vector<CPoint> take() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        if (err) std::rethrow_exception(err);
        return std::move(C);
    }
    static hipStream_t masked_stream() { return bulk_masked_stream(); }   // k_context.hip: ONE CU-masked stream for the process
    void run() {
        try {
            {   // created inside the prover's session: give back the affinity the session narrowed (this thread waits on the GPU, it does not belong on the helpers' cores)
                cpu_set_t all; CPU_ZERO(&all); for (int i = 0; i < CPU_SETSIZE; i++) CPU_SET(i, &all);
                (void)sched_setaffinity(0, sizeof all, &all);
            }
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stage != 0; }); if (stage < 0) { done = true; cv.notify_all(); return; } }
            DevCtx &hc = DevCtx::get();                           // this thread's own context: its own MSM partials, point buffers, stream
            hipStream_t own = hc.stream, ms = masked_stream();
            struct Restore { DevCtx &c; hipStream_t s; ~Restore() { c.stream = s; } } restore{hc, own};
            if (ms) hc.stream = ms;
            const bool trace = getenv("OTTI_TRACE") != nullptr; const double t_go = now_ms();
            OTTI_HIP(hipStreamWaitEvent(hc.stream, ev, 0));
            dev_msm_rows(hc, *gens.dev, Z, R, R, rows, nullptr, nullptr, 0, MSM_COMPRESSED, nullptr, false);
            const double t_queued = now_ms();
            // (polled, not hipStreamSynchronize: a blocking wait in this thread was seen to hold up the proving thread's launches for as long as it lasted)
            if (!getenv("OTTI_DEREFS_SYNC")) while (hipStreamQuery(hc.stream) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(50));
            hc.sync();
            if (trace) fprintf(stderr, "[otti] rows ahead: %zu rows on %s: queued in %.3f ms, summed %.3f ms after the release\n", rows, ms ? "the CU-masked stream" : "an ordinary second stream (no CU mask)", t_queued - t_go, now_ms() - t_go);
            std::vector<CPoint> out(rows); memcpy(out.data(), hc.h_points, 32 * rows);
            std::lock_guard<std::mutex> lk(mu); C = std::move(out); done = true;
        } catch (...) {
            // whatever was queued on the (shared) masked stream may still be using this context's buffers: drain it before the context goes back to the pool
            if (hipStream_t ms = masked_stream()) (void)hipStreamSynchronize(ms);
            std::lock_guard<std::mutex> lk(mu); err = std::current_exception(); done = true;
        }
        cv.notify_all();
    }
};

// ================================================================================================ SNARK::prove
std::vector<uint8_t> snark_prove_gpu(Instance &I, CompComm &comm, const uint8_t *vars32, size_t nvars, const std::vector<Fr> &inputs, SnarkGens &g,
                                     const void *tlabel, size_t tlabel_len, const uint8_t *seed32, SnarkTimings *tm) {
    if (I.num_cons != comm.num_cons || I.num_vars != comm.num_vars || I.num_inputs != comm.num_inputs) throw Error(OTTI_ERR_BAD_ARG, "commitment belongs to another instance");
    DeviceWitness wit(I, vars32, nvars, inputs);                 // uploaded (and checked for canonical scalars) inside the call
    return snark_prove_resident(I, comm, wit, g, tlabel, tlabel_len, seed32, tm);
}
static std::vector<uint8_t> snark_prove_resident_once(Instance &I, CompComm &comm, DeviceWitness &wit, SnarkGens &g, const void *tlabel, size_t tlabel_len,
                                                      const uint8_t *seed32, SnarkTimings *tm, ShardComm *sh);
std::vector<uint8_t> snark_prove_resident(Instance &I, CompComm &comm, DeviceWitness &wit, SnarkGens &g, const void *tlabel, size_t tlabel_len,
                                          const uint8_t *seed32, SnarkTimings *tm, ShardComm *sh) {
    // (sharded: every rank of a node sees the same device conditions or none does — a rank that alone repeated its proof would leave the
    // others waiting in the exchange — so the persistent tail, which can time out, is not used at all there)
    if (sh) return snark_prove_resident_once(I, comm, wit, g, tlabel, tlabel_len, seed32, tm, sh);
    try { return snark_prove_resident_once(I, comm, wit, g, tlabel, tlabel_len, seed32, tm, nullptr); }
    catch (const TailTimeout &e) {
        // the persistent tail's grid was not resident as a whole (the GPU is shared, partitioned or CU-masked): a proof is a function of its
        // inputs and the tape seed, so proving again — with a launch per round from now on — gives the bytes the first attempt would have given
        if (g_tail_off.exchange(true)) throw;
        fprintf(stderr, "[otti] notice: %s; SNARK::prove repeats the proof with one launch per sum-check round, and this process keeps doing so\n", e.what());
        return snark_prove_resident_once(I, comm, wit, g, tlabel, tlabel_len, seed32, tm, nullptr);
    }
}
static std::vector<uint8_t> snark_prove_resident_once(Instance &I, CompComm &comm, DeviceWitness &wit, SnarkGens &g, const void *tlabel, size_t tlabel_len,
                                                      const uint8_t *seed32, SnarkTimings *tm, ShardComm *sh) {
    DevCtx &c = DevCtx::get();
    ActiveProof active;
    // first-use HIP objects are made BEFORE the session narrows this thread's affinity to its helpers' L3 group (pool.h hold_caller): a thread
    // the runtime starts while making them would inherit the narrowed mask for good
    if (!sh && c.num_cu >= 128) (void)bulk_masked_stream();
    if (!sh && getenv("OTTI_HASH_AHEAD")) (void)c.side_stream();
    SpinPool::Session pool_session;
    struct Sharded { bool was; explicit Sharded(bool on) : was(t_sharded_proof) { t_sharded_proof = on; } ~Sharded() { t_sharded_proof = was; } } sharded_scope(sh != nullptr);
    if (!comm.dec) throw Error(OTTI_ERR_BAD_ARG, "this computation commitment carries no decommitment (it was parsed from bytes): SNARK::prove needs the one SNARK::encode returned");
    if (I.num_cons != comm.num_cons || I.num_vars != comm.num_vars || I.num_inputs != comm.num_inputs) throw Error(OTTI_ERR_BAD_ARG, "commitment belongs to another instance");
    const double t_start = now_ms(); SnarkTimings T{}; double t0;
    const bool trace = getenv("OTTI_TRACE") != nullptr; double t_lap = now_ms();
    auto lap = [&](const char *what) { if (!trace) return; c.sync(); const double t = now_ms(); fprintf(stderr, "[otti] snark_prove %-34s %.3f ms\n", what, t - t_lap); t_lap = t; };
    ensure_gens_device(*g.eval);
    const DeviceDecomm &d = *comm.dec; const size_t N = d.N, M = d.M, H = N / 2;
    if (wit.z.n != 2 * I.num_vars || wit.inputs.size() != I.num_inputs) throw Error(OTTI_ERR_BAD_ARG, "witness belongs to another instance");
    Transcript tr(tlabel, tlabel_len);
    RandomTape tape(seed32);
    SnarkProof S;
    tr.append_protocol_name("Spartan SNARK proof");
    snark_append_comm(tr, comm);
    SnarkScratch &W = snark_workspace(c);
    struct Ptr { Fr *p; };                                    // (the buffers below used to be DevBufs of this proof; the code keeps reading x.p)
    const Ptr mem_rx{W.get(SS_MEM_RX, M)}, mem_ry{W.get(SS_MEM_RY, M)}, eqs{W.get(SS_EQS, 5 * 4096)}, derefs{W.get(SS_DEREFS, (size_t)8 * N)},
              partials{W.get(SS_PARTIALS, (size_t)3 * 2048 + 64)};             // 3 sums x at most 2048 workgroups per launch (k_snark.hip many_grid)
    auto drow = [&](int k) { return derefs.p + (size_t)k * N; };
    auto dcol = [&](int k) { return derefs.p + (size_t)(3 + k) * N; };
    const size_t nm = ilog2(M);
    auto equalized = [&](const std::vector<Fr> &r) {          // zeros in FRONT of the shorter point
        if (r.size() > nm) throw Error(OTTI_ERR_INTERNAL, "memory size does not match the evaluation point");
        std::vector<Fr> e(nm - r.size(), fr_zero()); e.insert(e.end(), r.begin(), r.end()); return e;
    };
    // dense.deref, row side: row_ops_val[k][i] = eq(rx)[row_addr[k][i]] — queued the moment rx is final (R1csHooks), and with it the row half
    // of the derefs commitment on the helper's stream (RowsAhead above); the column side follows after the R1CS proof
    const size_t rows_half = g.derefs.R ? 3 * N / g.derefs.R : 0;
    static const bool ahead_env = [] { const char *e = getenv("OTTI_DEREFS_AHEAD"); return !(e && e[0] == '0'); }();
    const bool ahead = !sh && ahead_env && c.armed_ok() && c.num_cu >= 128 && N >= ((size_t)1 << 14) && rows_half >= 64 && rows_half * g.derefs.R == 3 * N && 2 * rows_half <= g.derefs.L;
    std::unique_ptr<RowsAhead> rows_job;
    if (ahead) rows_job.reset(new RowsAhead(*g.eval, derefs.p, g.derefs.R, rows_half));
    bool rows_queued = false;
    auto queue_rows = [&](const std::vector<Fr> &rx_) {
        const std::vector<Fr> rxe_ = equalized(rx_);
        dev_eq_evals(c, rxe_.data(), nm, mem_rx.p, eqs.p);
        for (int k = 0; k < 3; k++) dev_gather(c, mem_rx.p, d.row_addr[k].p, drow(k), N);
        if (rows_job) rows_job->record(c.stream);
        rows_queued = true;
    };
    // the helper starts once the second sum-check is past its bandwidth-bound rounds: those want the whole chip (confined to the CUs the
    // helper leaves free they took 3.2 ms instead of 0.75), the short rounds after them and the evaluation proof do not
    R1csHooks hooks; hooks.on_rx = queue_rows; hooks.on_idle = [&] { if (rows_job) rows_job->release(); };
    { ProveTimings pt{}; r1cs_prove_device(I, wit, *g.sat, tr, tape, S.r1cs, pt, sh, ahead ? &hooks : nullptr); for (int k = 0; k < 6; k++) T.ms[k] = pt.ms[k]; }
    lap("r1cs proof");
    // inst.evaluate(rx, ry) is not computed by a sparse product of its own: M(rx, ry) = sum_i val_i eq(rx)[row_i] eq(ry)[col_i] is exactly
    // the dot product of the dereferenced vectors the evaluation proof needs anyway (its two halves are E.dotp_left / dotp_right below),
    // so the eq tables, the gathers and one launch of six sums come first and serve both.
    EvalProof &E = S.eval;
    t0 = now_ms();
    const std::vector<Fr> &rx = S.r1cs.rx, &ry = S.r1cs.ry;
    if (((size_t)1 << std::max(rx.size(), ry.size())) != M) throw Error(OTTI_ERR_INTERNAL, "memory size does not match the evaluation point");
    const std::vector<Fr> rxe = equalized(rx), rye = equalized(ry);
    if (!rows_queued) queue_rows(rx);
    if (rows_job) rows_job->release();
    dev_eq_evals(c, rye.data(), nm, mem_ry.p, eqs.p);
    // comb = merge(rows, cols), zero-padded to 8 N
    OTTI_HIP(hipMemsetAsync(derefs.p + 6 * N, 0, 2 * N * sizeof(Fr), c.stream));
    lap("allocations, eq tables");
    for (int k = 0; k < 3; k++) dev_gather(c, mem_ry.p, d.col_addr[k].p, dcol(k), N);
    std::vector<Fr> dotp_evals(6);
    {
        AbcList A; A.n = 6;
        for (int k = 0; k < 3; k++) for (int half = 0; half < 2; half++) { A.A[2 * k + half] = drow(k) + half * H; A.B[2 * k + half] = dcol(k) + half * H; A.C[2 * k + half] = d.part(4, k) + half * H; }
        dev_sum3(c, A, H, partials.p, kSumSlot + 40);
        c.sync();
        for (int i = 0; i < 6; i++) dotp_evals[i] = c.h_results[kSumSlot + 40 + i];
        for (int k = 0; k < 3; k++) S.inst_evals[k] = fr_add(dotp_evals[2 * k], dotp_evals[2 * k + 1]);
    }
    lap("deref gathers, inst.evaluate");
    tr.append_scalar("Ar_claim", S.inst_evals[0]); tr.append_scalar("Br_claim", S.inst_evals[1]); tr.append_scalar("Cr_claim", S.inst_evals[2]);
    // ---- R1CSEvalProof::prove -> SparseMatPolyEvalProof::prove
    tr.append_protocol_name("Sparse polynomial evaluation proof");
    if (rows_job) {
        // the column half (and nothing for the zero rows: the identity compresses to 32 zero bytes) on this stream, beside what is left of the helper's row half
        dev_msm_rows(c, *g.eval->dev, derefs.p + (size_t)3 * N, g.derefs.R, g.derefs.R, rows_half, nullptr, nullptr, 0, MSM_COMPRESSED, nullptr, false);
        c.sync();
        E.comm_derefs.assign(g.derefs.L, CPoint{});
        for (auto &z : E.comm_derefs) memset(z.b, 0, 32);
        memcpy(E.comm_derefs[rows_half].b, c.h_points, 32 * rows_half);
        const std::vector<CPoint> head = rows_job->take();
        memcpy(E.comm_derefs[0].b, head[0].b, 32 * rows_half);
        rows_job.reset();
    } else if (sh && sh->world() > 1) {
        // one proof over several GPUs: the commitment's rows are independent MSMs over the same generators, so every rank sums a block of
        // them and the 32-byte results are gathered (rank order = row order), as the witness commitment of the R1CS proof is.  Only the
        // rows that can be non-zero are dealt out (the polynomial ends in 2 N zeros: their rows are the identity, 32 zero bytes).
        const size_t Lr = g.derefs.L, Rr = g.derefs.R, nzr = std::min(Lr, (6 * N + Rr - 1) / Rr), world = (size_t)sh->world();
        const size_t per = (nzr + world - 1) / world, r0 = std::min(nzr, per * (size_t)sh->rank()), mine = std::min(per, nzr - r0);
        if (per * 32 > ShardComm::kSlotBytes) throw Error(OTTI_ERR_BAD_ARG, "derefs commitment: too many rows per rank for the exchange");
        ensure_gens_device(*g.eval);
        std::vector<CPoint> block(per); for (auto &z : block) memset(z.b, 0, 32);
        if (mine) {
            dev_msm_rows(c, *g.eval->dev, derefs.p + r0 * Rr, Rr, Rr, mine, nullptr, nullptr, 0, MSM_COMPRESSED, nullptr, false);
            c.sync();
            memcpy(block[0].b, c.h_points, 32 * mine);
        }
        std::vector<CPoint> all(per * world);
        sh->allgather(block.data(), per * 32, all.data());
        E.comm_derefs.assign(Lr, CPoint{});
        for (auto &z : E.comm_derefs) memset(z.b, 0, 32);
        memcpy(E.comm_derefs[0].b, all[0].b, 32 * nzr);              // rank k's block starts at row k * per: contiguous up to nzr
    } else E.comm_derefs = commit_poly(c, *g.eval, derefs.p, g.derefs, 0);      // values of eq tables: uniform field elements
    tr.append_message("derefs_commitment", "begin_derefs_commitment", 23);
    append_poly_commitment(tr, "comm_poly_row_col_ops_val", E.comm_derefs);
    tr.append_message("derefs_commitment", "end_derefs_commitment", 21);
    lap("derefs commitment");
    T.ms[6] = now_ms() - t0;

    t0 = now_ms();
    const std::vector<Fr> r_mem_check = tr.challenge_vector("challenge_r_hash", 2);
    // PolyEvalNetwork::new: hash layers -> product circuits.  ops: row reads A,B,C; row writes; col reads; col writes.  mem: row init, row audit, col init, col audit.
    // One proof over several GPUs: the product circuits (hash layer, product layers, the large rounds of the two batched sum-checks — the
    // bandwidth-bound part of this stage) are split by residue classes of the element index; circuits too small for that run on every rank alike.
    const int pcG = (sh && sh->world() > 1 && sh->world() <= 16 && N >= (size_t)64 * sh->world() && M >= (size_t)64 * sh->world()) ? sh->world() : 1;
    const int pcR = pcG > 1 ? sh->rank() : 0;
    ShardComm *const pc_sh = pcG > 1 ? sh : nullptr;
    Circuits ops, mem;
    ops.init(12, N, W, SS_OPS, pcG, pcR); mem.init(4, M, W, SS_MEMC, pcG, pcR);
    lap("circuit allocations");
    for (int k = 0; k < 3; k++) {
        dev_hash_ops(c, d.part(0, k), drow(k), d.part(1, k), ops.input(k), ops.input(3 + k), N, r_mem_check[0], r_mem_check[1], pcG, pcR);
        dev_hash_ops(c, d.part(2, k), dcol(k), d.part(3, k), ops.input(6 + k), ops.input(9 + k), N, r_mem_check[0], r_mem_check[1], pcG, pcR);
    }
    dev_hash_mem(c, mem_rx.p, d.comb_mem.p, mem.input(0), mem.input(1), M, r_mem_check[0], r_mem_check[1], pcG, pcR);
    dev_hash_mem(c, mem_ry.p, d.comb_mem.p + M, mem.input(2), mem.input(3), M, r_mem_check[0], r_mem_check[1], pcG, pcR);
    lap("hash layer kernels");
    ops.build(c); mem.build(c);
    if (pc_sh) { ops.gather_small(c, *pc_sh); mem.gather_small(c, *pc_sh); }
    if (pc_sh && trace) fprintf(stderr, "[otti] snark_prove product circuits split by residue classes over %d ranks (rank %d: %zu of %zu operations, %zu of %zu memory cells)\n", pcG, pcR, ops.n_dev, N, mem.n_dev, M);
    lap("product layers");
    // the dot-product circuits: halves of (row_ops_val, col_ops_val, val) per matrix — copies, because the sum-check folds them in place
    const Ptr dotp{W.get(SS_DOTP, (size_t)9 * N)};
    DotpTables D; D.len = H / (size_t)pcG; D.n = 6;
    for (int k = 0; k < 3; k++) {
        Fr *base = dotp.p + (size_t)3 * k * N;
        if (pc_sh) {                                              // this rank's residue class of every half: element e = element e G + rk of the half
            const Fr *src[3] = {drow(k), dcol(k), d.part(4, k)};
            for (int t = 0; t < 3; t++) for (int half = 0; half < 2; half++)
                dev_gather_strided(c, src[t] + half * H, (size_t)pcG, (size_t)pcR, base + (size_t)t * N + half * H, D.len);
        } else {
            OTTI_HIP(hipMemcpyAsync(base, drow(k), N * sizeof(Fr), hipMemcpyDeviceToDevice, c.stream));
            OTTI_HIP(hipMemcpyAsync(base + N, dcol(k), N * sizeof(Fr), hipMemcpyDeviceToDevice, c.stream));
            OTTI_HIP(hipMemcpyAsync(base + 2 * N, d.part(4, k), N * sizeof(Fr), hipMemcpyDeviceToDevice, c.stream));
        }
        for (int half = 0; half < 2; half++) { D.l[2 * k + half] = base + half * H; D.r[2 * k + half] = base + N + half * H; D.w[2 * k + half] = base + 2 * N + half * H; }
    }
    // PolyEvalNetworkProof::prove / ProductLayerProof::prove
    tr.append_protocol_name("Sparse polynomial evaluation proof");
    tr.append_protocol_name("Sparse polynomial product layer proof");
    std::vector<Fr> ops_evals(12), mem_evals(4);
    if (pc_sh) {   // circuit outputs, sharded: the top layers are on the host (Circuits::small)
        for (int i = 0; i < 12; i++) ops_evals[i] = fr_mul(ops.small[ops.nl - 1][i].first[0], ops.small[ops.nl - 1][i].second[0]);
        for (int i = 0; i < 4; i++) mem_evals[i] = fr_mul(mem.small[mem.nl - 1][i].first[0], mem.small[mem.nl - 1][i].second[0]);
    } else {   // circuit outputs (left * right of the top layer); the dot-product claims were summed above
        PtrList pick; pick.n = 0;
        for (int i = 0; i < 12; i++) { pick.p[pick.n++] = ops.left(i, ops.nl - 1); pick.p[pick.n++] = ops.right(i, ops.nl - 1); }
        for (int i = 0; i < 4; i++) { pick.p[pick.n++] = mem.left(i, mem.nl - 1); pick.p[pick.n++] = mem.right(i, mem.nl - 1); }
        dev_pick0(c, pick, kSumSlot);
        c.sync();
        for (int i = 0; i < 12; i++) ops_evals[i] = fr_mul(c.h_results[kSumSlot + 2 * i], c.h_results[kSumSlot + 2 * i + 1]);
        for (int i = 0; i < 4; i++) mem_evals[i] = fr_mul(c.h_results[kSumSlot + 24 + 2 * i], c.h_results[kSumSlot + 24 + 2 * i + 1]);
    }
    E.eval_row.init = mem_evals[0]; E.eval_row.audit = mem_evals[1]; E.eval_col.init = mem_evals[2]; E.eval_col.audit = mem_evals[3];
    for (int k = 0; k < 3; k++) { E.eval_row.read[k] = ops_evals[k]; E.eval_row.write[k] = ops_evals[3 + k]; E.eval_col.read[k] = ops_evals[6 + k]; E.eval_col.write[k] = ops_evals[9 + k]; }
    for (int side = 0; side < 2; side++) {
        const Evals4 &e = side ? E.eval_col : E.eval_row;
        tr.append_scalar(side ? "claim_col_eval_init" : "claim_row_eval_init", e.init);
        tr.append_scalars(side ? "claim_col_eval_read" : "claim_row_eval_read", e.read, 3);
        tr.append_scalars(side ? "claim_col_eval_write" : "claim_row_eval_write", e.write, 3);
        tr.append_scalar(side ? "claim_col_eval_audit" : "claim_row_eval_audit", e.audit);
    }
    for (int k = 0; k < 3; k++) {
        E.dotp_left[k] = dotp_evals[2 * k]; E.dotp_right[k] = dotp_evals[2 * k + 1];
        tr.append_scalar("claim_eval_dotp_left", E.dotp_left[k]); tr.append_scalar("claim_eval_dotp_right", E.dotp_right[k]);
        if (!fr_eq(fr_add(E.dotp_left[k], E.dotp_right[k]), S.inst_evals[k])) throw Error(OTTI_ERR_INTERNAL, "sparse polynomial evaluation does not match its dot-product circuits");
    }
    std::vector<Fr> rand_ops, rand_mem;
    // The hash layer's work that depends on rand_ops alone.  The two evaluation proofs' points are [3 or 4 challenges drawn later | rand_ops] and a
    // committed vector is one CHUNK of rows of its polynomial's matrix, so ONE pass over the 21 vectors gives both the chunk sums of the proofs' bounds
    // (dev_poly_bound_chunks: P_c[j] = sum_i' eq(rand_ops[0 .. rest))[i'] v_c[i' R + j]) and, as dot products of length R with eq(rand_ops[rest ..]),
    // the 21 evaluations v_c(rand_ops) themselves — instead of a pass for the evaluations (with an eq table of N elements) and one per bound (3.1 GB
    // -> 1.6 GB at 2^20: hash layer 3.11 -> 2.9 ms).  OTTI_HASH_FUSED=0: the separate passes as before.  OTTI_HASH_AHEAD=1: that pass queued on a second
    // stream when rand_ops comes out, beside the memory circuits' sum-check (hash layer 2.9 -> 2.6 ms, but those rounds lose 0.1-0.35 ms — more than
    // that inside bench.py's SNARK leg, whatever CU mask the second stream has: profiles/r4_hash_layer_ab.txt; off by default).
    static const bool hash_fused_env = [] { const char *e = getenv("OTTI_HASH_FUSED"); return !(e && e[0] == '0'); }();
    static const bool hash_ahead_env = [] { const char *e = getenv("OTTI_HASH_AHEAD"); return e && e[0] == '1'; }();
    const size_t lgN_ = ilog2(N);
    const bool hash_fused = hash_fused_env && g.derefs.num_vars == lgN_ + 3 && g.ops.num_vars == lgN_ + 4 && g.derefs.num_vars / 2 > 3 && g.ops.num_vars / 2 > 4 && g.derefs.L * g.derefs.R == 8 * N && g.ops.L * g.ops.R == 16 * N;
    const bool hash_ahead = hash_fused && hash_ahead_env && !sh && c.side_stream();
    Fr *chunks_d = nullptr, *chunks_o = nullptr;
    constexpr int kHashEvalSlot = kPcPreExportEnd;                        // result slots of the 21 evaluations (clear of the round sums and the exported tails)
    bool hash_evals_queued = false;
    // work queued on the second stream uses this context's buffers: an exception on the way to the hash layer must not let the context go back to the pool under it
    struct SideDrain { DevCtx &c; bool pending = false; ~SideDrain() { if (pending && c.ev_side) (void)hipEventSynchronize(c.ev_side); } } side_drain{c};
    auto queue_hash_evals = [&] {                               // on c.stream; false (chunks_* null) when the bounds' geometry does not allow the chunked form
        const size_t lv_d = g.derefs.num_vars / 2, lv_o = g.ops.num_vars / 2, rest_d = lv_d - 3, rest_o = lv_o - 4;
        Fr *tab_d = W.get(SS_AHEAD + 2, (size_t)1 << rest_d), *tab_o = W.get(SS_AHEAD + 3, (size_t)1 << rest_o), *rv_d = W.get(SS_AHEAD + 4, g.derefs.R), *rv_o = W.get(SS_AHEAD + 5, g.ops.R);
        Fr *bound = W.get(SS_PE0 + 10, 64 * std::max(g.derefs.R, g.ops.R));
        chunks_d = W.get(SS_AHEAD, 8 * g.derefs.R); chunks_o = W.get(SS_AHEAD + 1, 16 * g.ops.R);
        dev_eq_evals2(c, rand_ops.data(), rest_d, tab_d, rand_ops.data(), rest_o, tab_o, eqs.p);
        if (!dev_poly_bound_chunks(c, derefs.p, g.derefs.L, g.derefs.R, tab_d, (size_t)1 << rest_d, chunks_d, bound) ||
            !dev_poly_bound_chunks(c, d.comb_ops.p, g.ops.L, g.ops.R, tab_o, (size_t)1 << rest_o, chunks_o, bound)) { chunks_d = chunks_o = nullptr; return false; }
        dev_eq_evals2(c, rand_ops.data() + rest_d, lgN_ - rest_d, rv_d, rand_ops.data() + rest_o, lgN_ - rest_o, rv_o, eqs.p);
        PtrList Ld, Lo; Ld.n = 0; Lo.n = 0;
        for (int k = 0; k < 6; k++) Ld.p[Ld.n++] = chunks_d + (size_t)k * g.derefs.R;            // = drow(0..2), dcol(0..2)
        for (int k = 0; k < 15; k++) Lo.p[Lo.n++] = chunks_o + (size_t)k * g.ops.R;              // = d.part(p, k), p = 0..4
        dev_dot_many(c, rv_d, Ld, g.derefs.R, partials.p, kHashEvalSlot);
        dev_dot_many(c, rv_o, Lo, g.ops.R, partials.p, kHashEvalSlot + 6);
        return true;
    };
    lap("dotp copies, circuit outputs");
    {
        const Ptr pyr{W.get(SS_PYR, 2 * 8192)};
        E.proof_ops = pcbatch_prove(c, ops, ops_evals, &D, dotp_evals, tr, pyr.p, rand_ops, pc_sh);
        lap("batched proof: ops");
        // rand_ops is out.  OTTI_HASH_AHEAD=1: the hash layer's rand_ops-only work to a second stream now (after the memory circuits' first launches)
        const std::function<void()> queue_hash_ahead = [&] {
            hipStream_t own = c.stream;
            struct Restore { DevCtx &c; hipStream_t s; ~Restore() { c.stream = s; } } restore{c, own};
            c.stream = c.side_stream();
            hash_evals_queued = queue_hash_evals();
            OTTI_HIP(hipEventRecord(c.ev_side, c.stream));
            side_drain.pending = true;
        };
        E.proof_mem = pcbatch_prove(c, mem, mem_evals, nullptr, {}, tr, pyr.p, rand_mem, pc_sh, hash_ahead ? &queue_hash_ahead : nullptr);
        lap("batched proof: mem");
    }
    T.ms[7] = now_ms() - t0;

    // ---- HashLayerProof::prove((rand_mem, rand_ops))
    t0 = now_ms();
    tr.append_protocol_name("Sparse polynomial hash layer proof");
    {   // evaluations at rand_ops of the six dereferenced vectors and the fifteen committed ones, at rand_mem of the two audit vectors
        const Ptr Em{W.get(SS_EM, M)};
        int es = kSumSlot;
        if (hash_ahead) {                                       // queued on the second stream when rand_ops came out
            OTTI_HIP(hipEventSynchronize(c.ev_side));
            OTTI_HIP(hipStreamWaitEvent(c.stream, c.ev_side, 0));
            side_drain.pending = false;
        } else if (hash_fused) hash_evals_queued = queue_hash_evals();
        if (hash_evals_queued) es = kHashEvalSlot;
        else {
            const Ptr Eo{W.get(SS_EO, N)};
            dev_eq_evals(c, rand_ops.data(), rand_ops.size(), Eo.p, eqs.p);
            PtrList Lo; Lo.n = 0;
            for (int k = 0; k < 3; k++) Lo.p[Lo.n++] = drow(k);
            for (int k = 0; k < 3; k++) Lo.p[Lo.n++] = dcol(k);
            for (int p = 0; p < 5; p++) for (int k = 0; k < 3; k++) Lo.p[Lo.n++] = d.part(p, k);
            dev_dot_many(c, Eo.p, Lo, N, partials.p, kSumSlot + 8);
        }
        if (!hash_evals_queued) es = kSumSlot + 8;
        dev_eq_evals(c, rand_mem.data(), rand_mem.size(), Em.p, eqs.p);
        PtrList Lm; Lm.n = 2; Lm.p[0] = d.comb_mem.p; Lm.p[1] = d.comb_mem.p + M;
        dev_dot_many(c, Em.p, Lm, M, partials.p, kSumSlot);
        c.sync();                                               // one synchronise for the 21 + 2 evaluations
        for (int k = 0; k < 3; k++) {
            E.h_deref_row[k] = c.h_results[es + k]; E.h_deref_col[k] = c.h_results[es + 3 + k];
            E.h_row_addr[k] = c.h_results[es + 6 + k]; E.h_row_read_ts[k] = c.h_results[es + 9 + k];
            E.h_col_addr[k] = c.h_results[es + 12 + k]; E.h_col_read_ts[k] = c.h_results[es + 15 + k]; E.h_val[k] = c.h_results[es + 18 + k];
        }
        E.h_row_audit = c.h_results[kSumSlot]; E.h_col_audit = c.h_results[kSumSlot + 1];
    }
    lap("hash layer evaluations");
    auto joint = [&](std::vector<Fr> ev, const char *label_evals, const char *label_ch, const char *label_joint, const std::vector<Fr> &rand, std::vector<Fr> &r_joint) {
        tr.append_scalars(label_evals, ev.data(), ev.size());
        std::vector<Fr> ch = tr.challenge_vector(label_ch, ilog2(ev.size()));
        const Fr j = reduce_evals(ev, ch);
        r_joint = ch; r_joint.insert(r_joint.end(), rand.begin(), rand.end());
        tr.append_scalar(label_joint, j);
        return j;
    };
    {   // DerefsEvalProof::prove
        tr.append_protocol_name("Derefs evaluation proof");
        std::vector<Fr> ev(8, fr_zero()), rj;
        for (int k = 0; k < 3; k++) { ev[k] = E.h_deref_row[k]; ev[3 + k] = E.h_deref_col[k]; }
        const Fr j = joint(ev, "evals_ops_val", "challenge_combine_n_to_one", "joint_claim_eval", rand_ops, rj);
        E.pe_derefs = polyeval_prove_plain(c, *g.eval, g.derefs, derefs.p, rj, j, tr, tape, chunks_d, 3);
    }
    lap("polyeval derefs");
    {
        std::vector<Fr> ev(16, fr_zero()), rj;
        for (int k = 0; k < 3; k++) { ev[k] = E.h_row_addr[k]; ev[3 + k] = E.h_row_read_ts[k]; ev[6 + k] = E.h_col_addr[k]; ev[9 + k] = E.h_col_read_ts[k]; ev[12 + k] = E.h_val[k]; }
        const Fr j = joint(ev, "claim_evals_ops", "challenge_combine_n_to_one", "joint_claim_eval_ops", rand_ops, rj);
        E.pe_ops = polyeval_prove_plain(c, *g.eval, g.ops, d.comb_ops.p, rj, j, tr, tape, chunks_o, 4);
    }
    lap("polyeval ops");
    {
        std::vector<Fr> rj;
        const Fr j = joint({E.h_row_audit, E.h_col_audit}, "claim_evals_mem", "challenge_combine_two_to_one", "joint_claim_eval_mem", rand_mem, rj);
        E.pe_mem = polyeval_prove_plain(c, *g.eval, g.mem, d.comb_mem.p, rj, j, tr, tape);
    }
    lap("polyeval mem");
    T.ms[8] = now_ms() - t0;
    std::vector<uint8_t> out = S.serialize();
    T.ms[9] = now_ms() - t_start;
    OTTI_HIP(hipStreamSynchronize(c.stream));
    KStats::get().flush();
    if (tm) *tm = T;
    return out;
}

}  // namespace otti
