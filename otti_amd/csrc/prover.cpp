// NIZK::prove on the MI355X: orchestration of the device kernels around the sequential Fiat-Shamir transcript.
// Follows upstream libspartan `src/r1csproof.rs::R1CSProof::prove` / `src/lib.rs::NIZK::prove` step for step
// [RECALL; SURVEY.md App. A — /root/reference/Spartan is an empty submodule], reached from `spzk verify --nizk`
// [REF /root/reference/run.py:58, run.py:100].
//
// Data flow: witness, tables and generators stay in HBM for the whole proof.  Per sum-check round the device returns
// 2-3 field elements (96 B) through a pinned buffer; the host hashes, derives the challenge, and launches the next
// fused fold+evaluate kernel before finishing the round's O(1) sigma-protocol work, so the two overlap.
#include "device.h"
#include "pool.h"
#include "shard.h"
#include <chrono>
#include <mutex>

namespace otti {

constexpr size_t kHostTailBits = 5;           // sum-check tables of at most 2^5 elements (all ranks together) are finished on the host
constexpr int kTailSlot = 64;                 // where their elements land in the pinned result buffer
constexpr double kSparseWitness = 0.25;       // above this share of small witness values the commitment uses the work-list MSM variant

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Window width c of the fixed-base table.  Every bit of c removes additions from every MSM of every proof (W = floor(253/c)+1 per
// scalar) and doubles the table; the table is built once per generator set and HBM is 288 GB, so take the widest window whose table
// fits a budget (default 200 GiB, two thirds of the card: c = 17 for R = 1024 (96.8 GB) and R = 2048 (193 GB), 16 for R = 4096 (206 GB); round 3 ran
// with 128 GiB and c <= 16: 16 / 16 / 15).  Two wide tables do not fit one card: the one built second takes the next narrower window that
// fits (ensure_gens_device), and otti_gens_release_device makes room.  Measured with the wider tables (bench.py --sweep, four timed proofs
// after two warm-ups): 2^22 6.95-7.0 -> 6.6 ms, 2^24 20.8-21.1 -> 20.2 ms.  OTTI_MSM_WINDOW pins c;
// OTTI_MSM_TABLE_GB changes the budget (one-shot callers such as spzk pick a small table: building it costs more than it saves).
int device_window_bits(size_t nbases) {
    if (const char *e = getenv("OTTI_MSM_WINDOW")) { int c = atoi(e); if (c >= 4 && c <= 17) return c; }
    double budget_gb = 200.0;
    if (const char *e = getenv("OTTI_MSM_TABLE_GB")) { double v = atof(e); if (v > 0) budget_gb = v; }
    int best = 8;
    const int widest = nbases < 256 ? 12 : 17;             // tiny instances (R < 256) are launch-bound whatever the window: keep their tables small
    for (int c = 8; c <= widest; c++) {
        const double W = 253 / c + 1, bytes = (double)nbases * W * (double)((size_t)1 << (c - 1)) * sizeof(TabEntry);
        if (bytes <= budget_gb * 1073741824.0) best = c;
    }
    return best;
}

static std::mutex &device_objects_mu() { static std::mutex m; return m; }   // lazily built HBM objects are shared by all prover threads
void ensure_instance_device(Instance &I) { std::lock_guard<std::mutex> lk(device_objects_mu()); if (!I.dev) I.dev = upload_instance(I); }
void ensure_gens_device(Gens &g) {
    std::lock_guard<std::mutex> lk(device_objects_mu());
    if (g.dev) return;
    // the widest window the budget allows; if HBM is short right now (other tenants of the GPU, other generator sets), a narrower one
    const int want = device_window_bits(g.R + 2);
    for (int c = want;; c--) {
        try {
            g.dev = build_device_gens(g, c);
            // a narrower table means more additions in every MSM of every proof: say so once (otti_gens_table_info reports the width in use)
            if (c != want) fprintf(stderr, "[otti] notice: HBM is short: generator window table built with c = %d instead of %d (%zu generators); proofs are unchanged, MSMs slower\n", c, want, g.R + 2);
            return;
        }
        catch (const OutOfDeviceMemory &) { if (c - 1 < 8 || getenv("OTTI_MSM_WINDOW")) throw; }
    }
}
// frees the window table (the next ensure_gens_device builds it again): a caller that needs the HBM for another generator set's table
void release_gens_device(Gens &g) { std::lock_guard<std::mutex> lk(device_objects_mu()); g.dev.reset(); }
void ensure_device_objects(Instance &I, Gens &g) { ensure_instance_device(I); ensure_gens_device(g); }

// the verifiers' fixed-base sums (spartan.h g_fixed_base_msm_hook): one row over the resident table; never builds a table for it
static bool fixed_base_msm_on_device(const Gens &g, const Fr *s, size_t n, Pt &out) {
    if (!g.dev || n < 256 || n > g.R) return false;
    try {
        DevCtx &c = DevCtx::get();
        DevBuf<Fr> d(n);
        OTTI_HIP(hipMemcpyAsync(d.p, s, n * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
        c.ensure_points(1, std::max<size_t>(1, n / 64));
        const unsigned long long tk = dev_msm_rows(c, *g.dev, d.p, n, n, 1, nullptr, nullptr, 0, MSM_RAW);
        if (tk) c.wait_ticket(tk);                               // fused launch: the extended row sum is mailed to pinned memory
        OTTI_HIP(hipStreamSynchronize(c.stream));
        c.pending_host_encode = 0;                               // the sum is wanted as a point, not compressed
        out = c.h_pts[0];
        return true;
    } catch (const Error &) { return false; }
}
// the verifiers' variable-base sums over decompressed row commitments (spartan.h g_row_sum_*_hook): decompression starts when the
// commitments are known, the LDS-bucket Pippenger (k_msm.hip k_msm_var) runs when the scalars are, the host recombines the windows
struct RowSumJob { DevCtx *c; RowSumSlot *slot; size_t n; };
static RowSumJob *row_sum_begin_on_device(const CPoint *C, size_t n) {
    if (n < 256 || n > ((size_t)1 << 20)) return nullptr;        // a few points: the host's own Pippenger is faster than two launches
    try {
        DevCtx &c = DevCtx::get();
        RowSumSlot *slot = nullptr;
        for (RowSumSlot *s : c.row_slots) if (!s->busy) { slot = s; break; }
        if (!slot) { if (c.row_slots.size() >= 8) return nullptr; c.row_slots.push_back(new RowSumSlot()); slot = c.row_slots.back(); }
        if (slot->comp.n < 32 * n) { slot->comp.alloc(32 * n); slot->pts.alloc(n); slot->sc.alloc(n); }
        if (!slot->bad.p) slot->bad.alloc(1);
        static_assert(sizeof(CPoint) == 32, "compressed points are uploaded as they lie in the proof");
        OTTI_HIP(hipMemcpyAsync(slot->comp.p, C, 32 * n, hipMemcpyHostToDevice, c.stream));
        OTTI_HIP(hipMemsetAsync(slot->bad.p, 0, sizeof(unsigned), c.stream));
        dev_decode_niels(c, slot->comp.p, n, slot->pts.p, slot->bad.p);
        slot->busy = true;
        return new RowSumJob{&c, slot, n};
    } catch (const Error &) { return nullptr; }
}
static int row_sum_finish_on_device(RowSumJob *job, const Fr *s, Pt &out) {
    std::unique_ptr<RowSumJob> own(job);
    struct Free { RowSumSlot *s; ~Free() { s->busy = false; } } release{job->slot};
    DevCtx &c = *job->c; RowSumSlot &S = *job->slot; const size_t n = job->n;
    try {
        if (!s) { OTTI_HIP(hipStreamSynchronize(c.stream)); return 0; }       // dropped: the decompression launch must not outlive the slot's lease
        if (&c != &DevCtx::get()) throw Error(OTTI_ERR_INTERNAL, "row sum finished on another thread than it was begun on");
        OTTI_HIP(hipMemcpyAsync(S.sc.p, s, n * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
        const size_t cap = 64 * ((n + 2047) / 2048);             // at most 51 windows (c = 5) x splits
        if (S.out.n < cap) S.out.alloc(cap);
        int W = 0, splits = 0;
        const int cbits = dev_msm_var(c, S.pts.p, S.sc.p, n, S.out.p, S.out.n, &W, &splits);
        std::vector<Pt> part((size_t)W * splits); unsigned bad = 0;
        OTTI_HIP(hipMemcpyAsync(part.data(), S.out.p, part.size() * sizeof(Pt), hipMemcpyDeviceToHost, c.stream));
        OTTI_HIP(hipMemcpyAsync(&bad, S.bad.p, sizeof bad, hipMemcpyDeviceToHost, c.stream));
        OTTI_HIP(hipStreamSynchronize(c.stream));
        if (bad) return OTTI_ERR_VERIFY_DECOMPRESS;
        // sum_w 2^(c w) * (sum over splits): Horner from the top window, c doublings per step
        Pt acc = pt_identity();
        for (int w = W - 1; w >= 0; w--) {
            if (w != W - 1) for (int k = 0; k < cbits; k++) acc = pt_dbl(acc);
            for (int sp = 0; sp < splits; sp++) acc = pt_add(acc, part[(size_t)w * splits + sp]);
        }
        out = acc;
        return 0;
    } catch (const Error &) { (void)hipStreamSynchronize(c.stream); return -1; }     // the caller falls back to the host cores
}
static const bool g_hook_registered = [] {
    g_fixed_base_msm_hook = fixed_base_msm_on_device; g_row_sum_begin_hook = row_sum_begin_on_device; g_row_sum_finish_hook = row_sum_finish_on_device;
    return true;
}();

DeviceWitness::DeviceWitness(const Instance &I, const std::vector<Fr> &vars_padded, const std::vector<Fr> &inputs_) : inputs(inputs_) {
    DevCtx &c = DevCtx::get();
    if (vars_padded.size() != I.num_vars) throw Error(OTTI_ERR_INVALID_NUM_VARS, "witness length != padded num_vars");
    z.alloc(2 * I.num_vars);
    // z = vars || 1 || inputs || 0...   (r1csproof.rs: "append input to variables to create a single vector z")
    std::vector<Fr> tail(I.num_vars, fr_zero());
    tail[0] = fr_one();
    for (size_t i = 0; i < inputs.size(); i++) tail[1 + i] = inputs[i];
    OTTI_HIP(hipMemcpyAsync(z.p, vars_padded.data(), I.num_vars * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
    OTTI_HIP(hipMemcpyAsync(z.p + I.num_vars, tail.data(), I.num_vars * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
    c.sync();
    small_fraction = dev_small_fraction(c, z.p, I.num_vars);
}

DeviceWitness::DeviceWitness(const Instance &I, const uint8_t *vars32, size_t nvars, const std::vector<Fr> &inputs_) : inputs(inputs_) {
    DevCtx &c = DevCtx::get();
    const size_t V = I.num_vars;
    if (nvars > V) throw Error(OTTI_ERR_INVALID_NUM_VARS, "more variables than the instance has");
    if (inputs.size() != I.num_inputs) throw Error(OTTI_ERR_INVALID_NUM_INPUTS, "wrong number of inputs");
    z.alloc(2 * V);
    OTTI_HIP(hipMemsetAsync(z.p, 0, 2 * V * sizeof(Fr), c.stream));
    if (nvars) OTTI_HIP(hipMemcpyAsync(z.p, vars32, nvars * 32, hipMemcpyHostToDevice, c.stream));
    std::vector<Fr> tail(1 + inputs.size()); tail[0] = fr_one();
    for (size_t i = 0; i < inputs.size(); i++) tail[1 + i] = inputs[i];
    OTTI_HIP(hipMemcpyAsync(z.p + V, tail.data(), tail.size() * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
    size_t n_small = 0;
    if (dev_witness_ingest(c, z.p, nvars, &n_small)) throw Error(OTTI_ERR_INVALID_SCALAR, "non-canonical scalar in assignment");   // synchronises: `tail` and the caller's buffer are free again
    small_fraction = V ? (double)(n_small + (V - nvars)) / (double)V : 0.0;        // the padding zeros count as small
}

namespace {
// Everything the rounds of BOTH sum-checks need that depends on the random tape alone, as ONE batched fixed-base MSM: for each round
// the four points delta_j = commit(d_vec_j, r_delta_j), blinds_poly[j]*h_n, blinds_evals[j]*h_1, r_beta_j*h_1 (extended, not
// compressed).  Launched right behind the witness commitment (the second sum-check's tape values are read ahead with a
// RandomTape::Cursor) and collected when the first sum-check starts.
struct RoundPointsJob { size_t n1 = 0, n2 = 0; std::vector<Fr> scalars; };   // scalars: host source of an in-flight async copy
RoundPointsJob precompute_round_points_launch(DevCtx &c, const DeviceGens &DG, const Gens &g, const SumcheckState &s1, const SumcheckState &s2, DevBuf<Fr> &scratch) {
    RoundPointsJob job; job.n1 = s1.pre.size(); job.n2 = s2.pre.size();
    const size_t n = job.n1 + job.n2;
    if (4 * n > kHostPtsCap) throw Error(OTTI_ERR_INTERNAL, "too many sum-check rounds");
    std::vector<uint32_t> bases;                                      // the generator-stream indices either sum-check touches (6 of them)
    auto col = [&](uint32_t idx) { for (size_t i = 0; i < bases.size(); i++) if (bases[i] == idx) return i; bases.push_back(idx); return bases.size() - 1; };
    struct Cols { size_t G[4], h, h1; } cs[2];
    const GensView *gv[2] = {&g.sc_4, &g.sc_3}; const size_t ne[2] = {4, 3};
    for (int k = 0; k < 2; k++) { for (size_t i = 0; i < ne[k]; i++) cs[k].G[i] = col(gv[k]->G[i]); cs[k].h = col(gv[k]->h); cs[k].h1 = col(g.sc_1.h); }
    const size_t nb = bases.size();
    if (nb > 8) throw Error(OTTI_ERR_INTERNAL, "sum-check generators do not share one short base list");
    std::vector<Fr> sc(4 * n * nb, fr_zero());
    if (scratch.n < sc.size()) scratch.alloc(sc.size());
    const SumcheckState *st[2] = {&s1, &s2}; size_t row = 0;
    for (int k = 0; k < 2; k++)
        for (size_t j = 0; j < st[k]->pre.size(); j++, row += 4) {
            Fr *r0 = &sc[(row + 0) * nb], *r1 = &sc[(row + 1) * nb], *r2 = &sc[(row + 2) * nb], *r3 = &sc[(row + 3) * nb];
            for (size_t i = 0; i < ne[k]; i++) r0[cs[k].G[i]] = st[k]->pre[j].d[i];
            r0[cs[k].h] = st[k]->pre[j].r_delta;
            r1[cs[k].h] = st[k]->blinds_poly[j];
            r2[cs[k].h1] = st[k]->blinds_evals[j];
            r3[cs[k].h1] = st[k]->pre[j].r_beta;
        }
    OTTI_HIP(hipMemcpyAsync(scratch.p, sc.data(), sc.size() * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
    dev_msm_rows(c, DG, nullptr, 0, 0, 4 * n, scratch.p, bases.data(), nb, MSM_RAW);
    OTTI_HIP(hipEventRecord(c.ev1, c.stream));
    job.scalars = std::move(sc);                                      // must outlive the copy queued above
    return job;
}
void precompute_round_points_collect(DevCtx &c, const RoundPointsJob &job, SumcheckState &s1, SumcheckState &s2) {
    OTTI_HIP(hipEventSynchronize(c.ev1));
    SumcheckState *st[2] = {&s1, &s2}; size_t row = 0;
    for (int k = 0; k < 2; k++)
        for (auto &p : st[k]->pre) { p.delta = c.h_pts[row]; p.bp_h = c.h_pts[row + 1]; p.be_h = c.h_pts[row + 2]; p.rb_h = c.h_pts[row + 3]; p.to_fe(); row += 4; }
    // compress the deltas now, off the per-round path, striped over the helper threads
    std::vector<RoundPre *> all; for (int k = 0; k < 2; k++) for (auto &p : st[k]->pre) all.push_back(&p);
    SpinPool &pool = SpinPool::get(); const int nt = pool.workers() + 1;
    std::vector<std::function<void()>> tasks(nt);
    for (int t = 0; t < nt; t++) tasks[t] = [&, t] { for (size_t j = t; j < all.size(); j += nt) pt_encode(all[j]->delta_c.b, all[j]->delta); };
    pool.parallel(tasks.data(), nt);
}
// HBM working set of one proof; kept across proofs of the same shape (hipMalloc/hipFree of ~0.5 GB costs more than a sum-check)
struct ProofScratch {
    size_t N = 0, V = 0;
    DevBuf<Fr> T[4];          // eq(tau), Az, Bz, Cz  (N each)
    DevBuf<Fr> zw, ABC;       // phase-two working tables (2V each)
    DevBuf<Fr> eqs;           // eq-table scratch (5 * 4096)
    DevBuf<Fr> pyr;           // the two eq pyramids of phase one: lo at [0, 8191), hi at [8192, 8192 + 16383)
    DevBuf<Fr> blinds, Lv, Rv, LZ, a, s, b2, s2, rows, extras, bound_scratch, pre;
    void reserve(size_t n, size_t v, size_t Lsz, size_t Rsz, size_t lgR) {
        if (n == N && v == V) return;
        for (auto &t : T) t.alloc(n);
        zw.alloc(2 * v); ABC.alloc(2 * v); eqs.alloc(5 * 4096); pyr.alloc(8192 + 16384);
        blinds.alloc(Lsz); Lv.alloc(Lsz); Rv.alloc(Rsz); LZ.alloc(Rsz); a.alloc(Rsz); s.alloc(Rsz); b2.alloc(Rsz); s2.alloc(Rsz); rows.alloc(2 * Rsz);
        extras.alloc(4 * (lgR + 1)); bound_scratch.alloc(64 * Rsz);
        N = n; V = v;
    }
};
}  // namespace
struct Scratch : ProofScratch {};                                // the type DevCtx::scratch points to (device.h)
namespace {
ProofScratch &workspace(DevCtx &c) { if (!c.scratch) c.scratch = new Scratch(); return *c.scratch; }   // one per context, kept across proofs

inline CPoint point_at(const DevCtx &c, size_t i) { CPoint p; memcpy(p.b, c.h_points + 32 * i, 32); return p; }
}  // namespace

// R1CSInstance::evaluate on the device, in the calling thread's proof workspace (nothing of a proof is live when a verifier or the SNARK
// prover's closing step calls this; allocating 160 MB for it per call cost more than the evaluation)
// begin queues the kernels on the calling thread's stream and returns; finish waits for them.  A verifier begins before it walks the
// sum-check rounds (the evaluation point is in the proof; the transcript's own challenges are compared with it at the end) and
// collects the three values where the closing equality needs them, ~2 ms later.
void instance_evaluate_begin(Instance &I, const std::vector<Fr> &rx, const std::vector<Fr> &ry) {
    DevCtx &c = DevCtx::get();
    ensure_instance_device(I);
    const size_t N = I.num_cons, V = I.num_vars, V2 = 2 * V;
    if (((size_t)1 << rx.size()) != N || ((size_t)1 << ry.size()) != V2) throw Error(OTTI_ERR_VERIFY_INTERNAL, "challenge vector lengths do not match the instance");
    ProofScratch &S = workspace(c);
    { const size_t ell = ilog2(V), Lsz = (size_t)1 << (ell / 2), Rsz = (size_t)1 << (ell - ell / 2); S.reserve(N, V, Lsz, Rsz, ilog2(Rsz)); }
    Fr *ex = S.T[0].p, *ey = S.zw.p, *Mz[3] = {S.T[1].p, S.T[2].p, S.T[3].p};
    dev_eq_evals(c, rx.data(), rx.size(), ex, S.eqs.p);
    dev_eq_evals(c, ry.data(), ry.size(), ey, S.eqs.p);
    dev_spmv3(c, I.dev->by_row, ey, Mz[0], Mz[1], Mz[2], false, nullptr);
    for (int k = 0; k < 3; k++) dev_dot(c, ex, Mz[k], N, kInstEvalSlot + k);
}
void instance_evaluate_finish(Fr out[3]) {
    DevCtx &c = DevCtx::get();
    c.sync();
    for (int k = 0; k < 3; k++) out[k] = c.h_results[kInstEvalSlot + k];
}
void instance_evaluate_gpu(Instance &I, const std::vector<Fr> &rx, const std::vector<Fr> &ry, Fr out[3]) {
    instance_evaluate_begin(I, rx, ry);
    instance_evaluate_finish(out);
}


// nizk/mod.rs DotProductProofLog::prove on device vectors: x = LZ (the bound polynomial row, R elements) against a = Rv, over the
// generators gens_n = P[0..R) / gens_1 of the stream `g` was derived from (PcView: stream indices).  Cx, the bullet-reduction rounds
// (on the ORIGINAL generators, k_msm.hip) and delta are fixed-base MSM launches; y = <x, a> is either given or read from result slot 12
// (a dev_dot queued by the caller).  Buffers are R elements each (rows: 2R, extras: 4 (lgR + 1)); LZ and Rv are consumed.
DotProductProofLog dplog_prove_device(DevCtx &c, const DeviceGens &DG, const Gens &g, const PcView &v, const PeBufs &B, const Fr &LZ_blind, const Fr *y_known,
                                      const Fr &blind_y, CPoint &Cy_out, Transcript &tr, RandomTape &tape) {
    const size_t Rsz = v.R, lgR = ilog2(Rsz);
    DotProductProofLog pf;
    tr.append_protocol_name("dot product proof (log)");
    Fr d = tape.random_scalar("d"), r_delta = tape.random_scalar("r_delta"), r_beta = tape.random_scalar("r_delta");   // sic: upstream reuses the label
    std::vector<Fr> bv1 = tape.random_vector("blinds_vec_1", 2 * lgR), bv2 = tape.random_vector("blinds_vec_2", 2 * lgR);
    unsigned long long tk_cx = 0;
    {   // Cx = commit(LZ, LZ_blind) over gens_n
        OTTI_HIP(hipMemcpyAsync(B.extras, &LZ_blind, sizeof(Fr), hipMemcpyHostToDevice, c.stream));
        uint32_t hb = v.h_n;
        tk_cx = dev_msm_rows(c, DG, B.LZ, Rsz, Rsz, 1, B.extras, &hb, 1);
    }
    c.wait_points(tk_cx);
    const Fr y = y_known ? *y_known : c.h_results[12];
    CPoint Cx = point_at(c, 0);
    tr.append_point("Cx", Cx.b);
    { Term t2[2] = {{v.g1, y}, {v.h1, blind_y}}; g.commit_terms_c(Cy_out.b, t2, 2); }
    tr.append_point("Cy", Cy_out.b);
    Fr blind_fin = fr_add(LZ_blind, blind_y);
    // BulletReductionProof::prove on the original generators (see k_msm.hip)
    std::vector<Fr> ex(4 * (lgR + 1), fr_zero());
    for (size_t k = 0; k < lgR; k++) { ex[4 * k + 1] = bv1[k]; ex[4 * k + 3] = bv2[k]; }
    ex[4 * lgR] = r_delta;                                            // delta's blind rides along (its slot is past every round's four)
    OTTI_HIP(hipMemcpyAsync(B.extras, ex.data(), ex.size() * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
    // round state ping-pongs between two buffer sets: launch k reads set k&1 and writes the folded state to set (k+1)&1
    Fr *abuf[2] = {B.LZ, B.a}, *bbuf[2] = {B.Rv, B.b2}, *sbuf[2] = {B.s, B.s2};
    dev_fill_one(c, sbuf[0], Rsz);
    const uint32_t qh[2] = {v.g1, v.h_n};
    Fr u = fr_zero(), ui = fr_zero();
    // round k works on vectors of length Rsz >> k; rounds k >= 1 first fold by (u_{k-1}, 1/u_{k-1}).  Armed (device.h), round k + 1 is
    // queued while round k runs and starts when the host publishes the challenge.
    const size_t n_rounds = lgR;
    const bool arm_ok = c.armed_ok() && !getenv("OTTI_MSM_STAMPS");
    struct Release { DevCtx &c; ~Release() { c.go_abort(); } } release{c};
    std::vector<unsigned long long> tks(n_rounds + 1, 0);
    auto launch_round = [&](size_t k, bool armed) {
        const int in = (int)(k & 1), out = in ^ 1;
        tks[k] = dev_bullet_round(c, DG, Rsz, Rsz >> k, k != 0, u, ui, abuf[in], bbuf[in], sbuf[in], abuf[out], bbuf[out], sbuf[out], B.extras + 4 * k, qh, armed);
    };
    if (n_rounds) launch_round(0, false);
    const bool arm = arm_ok && n_rounds > 1 && tks[0] != 0;      // only fused launches (results by mailbox, no stream synchronise behind which an armed kernel would wait for the host)
    if (arm) launch_round(1, true);
    size_t round = 0;
    for (; round < n_rounds; round++) {
        if (arm) c.pending_host_encode = 2;                      // the launch queued ahead has already re-armed the counter once; this round's L and R are still to be compressed
        c.wait_points(tks[round]);
        CPoint Lp = point_at(c, 0), Rp = point_at(c, 1);
        tr.append_point("L", Lp.b); tr.append_point("R", Rp.b);
        pf.L_vec.push_back(Lp); pf.R_vec.push_back(Rp);
        u = tr.challenge_scalar("u"); ui = fr_inv_fast(u);                 // (hostfast.h: 1.5 us instead of 8.8, once per round on the sequential path)
        if (round + 1 < n_rounds) {
            if (arm) { const Fr v4[4] = {u, ui, fr_to_raw(u), fr_to_raw(ui)}; c.go(v4, 4); if (round + 2 < n_rounds) launch_round(round + 2, true); }
            else launch_round(round + 1, false);
        }
        blind_fin = fr_add(blind_fin, fr_add(fr_mul(fr_mul(bv1[round], u), u), fr_mul(fr_mul(bv2[round], ui), ui)));
    }
    // last fold (length 2 -> 1) in place on the current set; s gets its final coefficients
    Fr *afin = abuf[round & 1], *bvec = bbuf[round & 1], *sfin = sbuf[round & 1];
    // delta = d * g_hat + r_delta * h with g_hat = sum_j s[j] P[j]: the last fold, the folded a and b for the host (slots 13, 14) and d s in one launch
    if (round) dev_bullet_finish(c, afin, bvec, sfin, Rsz, u, ui, d, B.rows, 13);
    else { dev_fetch(c, afin, 13, 1); dev_fetch(c, bvec, 14, 1); dev_scale(c, sfin, d, B.rows, Rsz); }
    unsigned long long tk_delta;
    { uint32_t hb = v.h1; tk_delta = dev_msm_rows(c, DG, B.rows, Rsz, Rsz, 1, B.extras + 4 * lgR, &hb, 1); }
    c.wait_points(tk_delta);
    const Fr x_hat = c.h_results[13], a_hat = c.h_results[14];
    pf.delta = point_at(c, 0);
    tr.append_point("delta", pf.delta.b);
    { Term t2[2] = {{v.g1, d}, {v.h1, r_beta}}; g.commit_terms_c(pf.beta.b, t2, 2); }
    tr.append_point("beta", pf.beta.b);
    Fr ch = tr.challenge_scalar("c");
    Fr y_hat = fr_mul(x_hat, a_hat);
    pf.z1 = fr_add(d, fr_mul(ch, y_hat));
    pf.z2 = fr_add(fr_mul(a_hat, fr_add(fr_mul(ch, blind_fin), r_beta)), r_delta);
    return pf;
}

// The last log2(g) rounds of a sharded sum-check: every rank holds the same g-element tables on the host (one element came from each
// rank) and plays the rounds there.  Same arithmetic as k_sc_cubic_eval / k_sc_quad_eval / k_fold_top.
static void host_cubic_evals(const std::vector<Fr> T[4], Fr e[3]) {
    const size_t h = T[0].size() / 2;
    e[0] = e[1] = e[2] = fr_zero();
    for (size_t i = 0; i < h; i++) {
        Fr lo[4], hi[4], p2[4], p3[4];
        for (int k = 0; k < 4; k++) { lo[k] = T[k][i]; hi[k] = T[k][h + i]; Fr d = fr_sub(hi[k], lo[k]); p2[k] = fr_add(hi[k], d); p3[k] = fr_add(p2[k], d); }
        e[0] = fr_add(e[0], fr_mul(lo[0], fr_sub(fr_mul(lo[1], lo[2]), lo[3])));
        e[1] = fr_add(e[1], fr_mul(p2[0], fr_sub(fr_mul(p2[1], p2[2]), p2[3])));
        e[2] = fr_add(e[2], fr_mul(p3[0], fr_sub(fr_mul(p3[1], p3[2]), p3[3])));
    }
}
static void host_quad_evals(const std::vector<Fr> T[2], Fr e[2]) {
    const size_t h = T[0].size() / 2;
    e[0] = e[1] = fr_zero();
    for (size_t i = 0; i < h; i++) {
        Fr a2 = fr_add(T[0][h + i], fr_sub(T[0][h + i], T[0][i])), b2 = fr_add(T[1][h + i], fr_sub(T[1][h + i], T[1][i]));
        e[0] = fr_add(e[0], fr_mul(T[0][i], T[1][i]));
        e[1] = fr_add(e[1], fr_mul(a2, b2));
    }
}
static void host_fold_top(std::vector<Fr> &t, const Fr &r) {
    const size_t h = t.size() / 2;
    for (size_t i = 0; i < h; i++) t[i] = fr_add(t[i], fr_mul(r, fr_sub(t[h + i], t[i])));
    t.resize(h);
}

// R1CSProof::prove on the device.  The transcript already carries the caller's protocol name (NIZK / SNARK) and whatever that caller
// appends before the satisfiability proof; P receives the proof and the challenges (rx, ry); T.ms[0..5] the stage times.
void r1cs_prove_device(Instance &I, DeviceWitness &wit, Gens &g, Transcript &tr, RandomTape &tape, NizkProof &P, ProveTimings &T, ShardComm *sh, const R1csHooks *hooks) {
    DevCtx &c = DevCtx::get();
    ensure_device_objects(I, g);
    const DeviceInstance &DI = *I.dev; const DeviceGens &DG = *g.dev;
    const size_t N = I.num_cons, V = I.num_vars, nrx = ilog2(N), nry = ilog2(2 * V);
    const size_t ell = ilog2(V), Lsz = (size_t)1 << (ell / 2), Rsz = (size_t)1 << (ell - ell / 2);
    if (g.num_vars_padded != V || g.R != Rsz) throw Error(OTTI_ERR_BAD_ARG, "generators were made for a different instance size");
    if (wit.inputs.size() != I.num_inputs) throw Error(OTTI_ERR_INVALID_NUM_INPUTS, "wrong number of inputs");
    // sharding: rank rk of G.  Tables are split by the LOW log2(G) index bits, so bound_poly_var_top's pairs (i, i + n/2)
    // stay on one GPU until a table is down to G elements; witness-matrix rows are split in G contiguous blocks.
    if (sh && sh->world() == 1) sh = nullptr;
    const size_t G = sh ? (size_t)sh->world() : 1, rk = sh ? (size_t)sh->rank() : 0, lgG = ilog2(G);
    const size_t Nl = N / G, V2l = 2 * V / G, Ll = Lsz / G;
    if (sh) {
        if (Nl < 2 || V2l < 2 || Ll < 1) throw Error(OTTI_ERR_BAD_ARG, "instance too small to shard over this many GPUs");
        std::lock_guard<std::mutex> lk(device_objects_mu());
        if (!I.shard || I.shard->rank != (int)rk || I.shard->world != (int)G) I.shard = upload_instance_shard(I, (int)rk, (int)G);
    }
    const DeviceCsrSet &rows_set = sh ? I.shard->by_row : DI.by_row, &cols_set = sh ? I.shard->by_col : DI.by_col;

    double t0;
    const size_t lgR = ilog2(Rsz);
    ProofScratch &S = workspace(c);
    S.reserve(N, V, Lsz, Rsz, lgR);
    c.ensure_points(std::max(Lsz, 4 * (nrx + nry)), 2 * std::max<size_t>(1, Rsz / 256));   // every MSM result buffer of this proof, before the first launch
    const Fr *d_vars = wit.z.p, *my_rows = d_vars + rk * Ll * Rsz;         // this rank's block of witness-matrix rows

    struct Release { DevCtx &c; ~Release() { c.go_abort(); } } release{c};      // an exception must not leave an armed kernel waiting
    SumcheckState early1, early2; RoundPointsJob round_points;       // tape-only parts of both sum-checks, started during polycommit
    tr.append_protocol_name("R1CS proof");

    // ---- polycommit: DensePolynomial::commit (K8).  The witness terms of every row are summed first (no host input needed);
    // meanwhile the host draws the whole random tape (its label sequence is known in advance); the blind terms are added last.
    t0 = now_ms();
    {
        // several proofs in flight in this process: the chip-filling launch goes to the process's CU-masked stream, so that the other
        // proofs' rounds (tens of workgroups each) find CUs whose registers no MSM workgroup holds; ordered against this context's
        // stream by events on both sides
        // (measured, tools/inflight_variants.sh, three runs each: 439-497 M constraints/s with the mask, 457-471 without — what held the in-flight
        // figure down was the helper threads' spinning, pool.h — so the mask is opt-in: OTTI_INFLIGHT_MASK=1)
        static const bool mask_env = [] { const char *e = getenv("OTTI_INFLIGHT_MASK"); return e && e[0] == '1'; }();
        hipStream_t bulk = (mask_env && !sh && ActiveProof::count() > 1) ? bulk_masked_stream() : nullptr;
        if (bulk) {
            hipStream_t own = c.stream;
            OTTI_HIP(hipEventRecord(c.ev1, own)); OTTI_HIP(hipStreamWaitEvent(bulk, c.ev1, 0));
            c.stream = bulk;
            struct Restore { DevCtx &c; hipStream_t s; ~Restore() { c.stream = s; } } restore{c, own};
            dev_msm_rows(c, DG, my_rows, Rsz, Rsz, Ll, nullptr, nullptr, 0, MSM_KEEP, nullptr, wit.small_fraction > kSparseWitness);
            OTTI_HIP(hipEventRecord(c.ev1, bulk)); OTTI_HIP(hipStreamWaitEvent(own, c.ev1, 0));
        } else dev_msm_rows(c, DG, my_rows, Rsz, Rsz, Ll, nullptr, nullptr, 0, MSM_KEEP, nullptr, wit.small_fraction > kSparseWitness);
    }
    size_t off_sc1 = 0, off_sc2 = 0;                                  // where the two sum-checks' draws sit in the prefetched tape
    {
        std::vector<std::pair<const char *, size_t>> sched;
        auto sumcheck_sched = [&](size_t rounds, size_t ne) {
            sched.push_back({"blinds_poly", rounds}); sched.push_back({"blinds_evals", rounds});
            for (size_t j = 0; j < rounds; j++) { sched.push_back({"d_vec", ne}); sched.push_back({"r_delta", 1}); sched.push_back({"r_beta", 1}); }
        };
        auto drawn = [&] { size_t n = 0; for (auto &e : sched) n += e.second; return n; };
        sched.push_back({"poly_blinds", Lsz});
        off_sc1 = drawn(); sumcheck_sched(nrx, 4);
        for (const char *l : {"Az_blind", "Bz_blind", "Cz_blind", "prod_Az_Bz_blind", "t1", "t2", "b1", "b2", "b3", "b4", "b5", "r"}) sched.push_back({l, 1});
        off_sc2 = drawn(); sumcheck_sched(nry, 3);
        sched.push_back({"blind_eval", 1}); sched.push_back({"d", 1}); sched.push_back({"r_delta", 2});
        sched.push_back({"blinds_vec_1", 2 * lgR}); sched.push_back({"blinds_vec_2", 2 * lgR}); sched.push_back({"r", 1});
        tape.prefetch(sched);
    }
    std::vector<Fr> blinds_vars = tape.random_vector("poly_blinds", Lsz);
    OTTI_HIP(hipMemcpyAsync(S.blinds.p, blinds_vars.data() + rk * Ll, Ll * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
    {
        uint32_t hbase = g.pc_n.h;
        dev_msm_rows(c, DG, nullptr, 0, 0, Ll, S.blinds.p, &hbase, 1, MSM_COMPRESSED, c.msm_keep.p);
        // Az, Bz, Cz do not depend on the transcript: queue them behind the commitment so that they run while the host hashes it
        OTTI_HIP(hipEventRecord(c.ev0, c.stream));
        dev_spmv3(c, rows_set, wit.z.p, S.T[1].p, S.T[2].p, S.T[3].p, false, nullptr);
        // ... and so do the blinding commitments of every sum-check round (tape-only): one batched launch for both sum-checks
        { RandomTape::Cursor c1(tape, off_sc1), c2(tape, off_sc2); sumcheck_draw_tape(early1, c1, nrx, 4); sumcheck_draw_tape(early2, c2, nry, 3); }
        round_points = precompute_round_points_launch(c, DG, g, early1, early2, S.pre);
        OTTI_HIP(hipEventSynchronize(c.ev0));
        c.encode_pending();
        P.comm_vars.resize(Lsz);
        static_assert(sizeof(CPoint) == 32, "CPoint is 32 packed bytes");
        if (sh) sh->allgather(c.h_points, Ll * 32, P.comm_vars.data());       // rank order = row-block order
        else for (size_t i = 0; i < Lsz; i++) P.comm_vars[i] = point_at(c, i);
        tr.append_message("poly_commitment", "poly_commitment_begin", 21);
        for (auto &cp : P.comm_vars) tr.append_point("poly_commitment_share", cp.b);
        tr.append_message("poly_commitment", "poly_commitment_end", 19);
    }
    T.ms[0] = now_ms() - t0;

    // ---- tau, eq(tau), Az/Bz/Cz (K2, K1)
    t0 = now_ms();
    std::vector<Fr> tau = tr.challenge_vector("challenge_tau", nrx);
    // eq(tau, .) is never materialised (k_sumcheck.hip, "phase one without the eq table"): two small pyramids of eq tables over the
    // local variables tau[0 .. s_loc) — the last n_lo of them, and the n_hi before those — give every round's E_j = eq(tau_{j+1..}, .).
    // Sharded: eq(tau, i'*G + rk) = eq(tau[0..s_loc), i') * eq(tau[s_loc..), rk)  (index bits are MSB-first), the second factor a scalar.
    const size_t s_loc = nrx - lgG, n_lo = std::min<size_t>(s_loc, 12), n_hi = s_loc - n_lo;
    Fr *pyr_lo = S.pyr.p, *pyr_hi = S.pyr.p + 8192;
    dev_eq_pyramid2(c, tau.data() + n_hi, n_lo, pyr_lo, tau.data(), n_hi, n_hi ? pyr_hi : nullptr);
    auto eq_src = [&](size_t m) {                                     // E = eq over the last m local variables
        EqSrc e;
        if (m <= n_lo) { e.hi = nullptr; e.lo = pyr_lo + (((size_t)1 << m) - 1); e.lo_bits = 0; }
        else { e.hi = pyr_hi + (((size_t)1 << (m - n_lo)) - 1); e.lo = pyr_lo + (((size_t)1 << n_lo) - 1); e.lo_bits = (int)n_lo; }
        return e;
    };
    const std::vector<Fr> eq_ranks = eq_evals_host(tau.data() + s_loc, lgG);   // eq(tau[s_loc..), rank) — {1} on one GPU
    c.sync();                                                         // multiply_vec was queued during polycommit; this stage is what is left of it
    T.ms[1] = now_ms() - t0;

    // ---- sum-check phase one (K3 + K4): comb = eq * (Az * Bz - Cz), claim 0
    t0 = now_ms();
    P.rx.resize(nrx);
    Fr blind_claim_postsc1, cj = fr_one();                            // cj = prod_{k<j} eq(tau_k, r_k): the eq table's scalar part so far
    {
        SumcheckState st;
        sumcheck_draw_tape(st, tape, nrx, 4);
        precompute_round_points_collect(c, round_points, early1, early2);
        st.pre = std::move(early1.pre);
        st.claim = fr_zero(); st.blind_claim = fr_zero();
        { Term t2[2] = {{g.sc_1.G[0], fr_zero()}, {g.sc_1.h, fr_zero()}}; g.commit_terms_c(st.comm_claim.b, t2, 2); }
        P.sc1.comm_polys.resize(nrx); P.sc1.comm_evals.resize(nrx); P.sc1.proofs.resize(nrx);
        // The device plays the rounds while the tables are long; once a table is down to a few dozen elements a round's kernel is
        // all launch + hand-off latency (~15 us more than the arithmetic costs on a host core), so the LAST rounds are played on the
        // host: the launch that produces the sums of round `ndev` also leaves the tables folded to T1 elements (per rank), they are
        // copied out behind it, and the host folds and sums from there on (same arithmetic: host_cubic_evals / host_fold_top).
        // Sharded, every rank does so identically on the T1 * G gathered elements.
        const size_t lgT1 = std::min(s_loc, kHostTailBits > lgG ? kHostTailBits - lgG : (size_t)0), T1 = (size_t)1 << lgT1, ndev = s_loc - lgT1;
        const size_t dsum = T1 >= 2 ? ndev + 1 : ndev;            // rounds whose sums the device delivers (a one-element local table has no pair left to sum)
        std::vector<Fr> tail1[4]; bool tail_built = false;
        auto fetch_tail = [&] { for (int k = 1; k < 4; k++) dev_fetch(c, S.T[k].p, kTailSlot + (k - 1) * (int)T1, T1); OTTI_HIP(hipEventRecord(c.ev0, c.stream)); };
        auto build_tail = [&] {                                    // the tables as the device left them: T1 elements per table and rank
            OTTI_HIP(hipEventSynchronize(c.ev0));
            const Fr *mine = &c.h_results[kTailSlot];
            std::vector<Fr> all;
            if (sh) { all.resize(3 * T1 * G); sh->allgather(mine, 3 * T1 * sizeof(Fr), all.data()); }
            for (int k = 1; k < 4; k++) {
                tail1[k].resize(T1 * G);
                for (size_t r = 0; r < G; r++) for (size_t i = 0; i < T1; i++) tail1[k][i * G + r] = (sh ? all.data() + 3 * T1 * r : mine)[(k - 1) * T1 + i];
            }
            tail1[0] = eq_evals_host(tau.data() + ndev, nrx - ndev);         // the eq table's remaining entries: c_j * eq(tau[ndev..), .)
            for (auto &x : tail1[0]) x = fr_mul(x, cj);
            tail_built = true;
        };
        // fold launch k (1..ndev) folds the tables of length Nl >> (k-1) by r_{k-1} and sums round k.  Armed (device.h), it is queued a
        // round ahead and starts the moment the host publishes r_{k-1}; otherwise it is launched once r_{k-1} is known.
        // Only launches of at most 64 workgroups are armed: those rounds are pure latency, and a waiting grid that small leaves the chip to
        // the other proofs in flight (and to the other ranks when several share one GPU).
        const bool arm_ok = c.armed_ok() && T1 >= 2;
        auto armed = [&](size_t k) { return arm_ok && k >= 1 && k <= ndev && (Nl >> (k - 1)) <= kArmMaxLen; };
        std::vector<unsigned long long> tk(ndev + 2, 0);
        auto fold_launch = [&](size_t k, const Fr *r) {
            const size_t len = Nl >> (k - 1);
            if (len >= 4) tk[k] = r ? dev_sc_cubic3_fold_eval(c, S.T[1].p, S.T[2].p, S.T[3].p, len, *r, eq_src(s_loc - k - 1), 0)
                                    : dev_sc_cubic3_fold_eval_armed(c, S.T[1].p, S.T[2].p, S.T[3].p, len, eq_src(s_loc - k - 1), 0);
            else for (int t = 1; t < 4; t++) dev_fold_top(c, S.T[t].p, len, *r);
            if (k == ndev) fetch_tail();
        };
        tk[0] = dev_sc_cubic3_eval(c, S.T[1].p, S.T[2].p, S.T[3].p, Nl, eq_src(s_loc - 1), 0);
        if (ndev == 0) fetch_tail(); else if (armed(1)) fold_launch(1, nullptr);
        const Fr one = fr_one();
        double tw = 0, tb = 0, tl = 0, tf = 0, ta;
        for (size_t j = 0; j < nrx; j++) {
            Fr e[3];
            ta = now_ms();
            if (j >= dsum && !tail_built) build_tail();
            if (j < dsum) {
                c.wait_ticket(tk[j]);
                e[0] = c.h_results[0]; e[1] = c.h_results[1]; e[2] = c.h_results[2];       // S_t = sum_i E_j[i] (Az_t Bz_t - Cz_t)[i], t = 0, 2, 3
                // per-round exchange: 96 bytes per rank (RCCL transport: the sums are packed into lanes, scaled, where the kernel left them in HBM)
                if (sh) { for (auto &x : e) x = fr_mul(x, eq_ranks[rk]); sh->allreduce_fr_device(c.results.p, &eq_ranks[rk], e, 3); }
                // e_t = c_j * w_t * S_t with w_t = (1 - tau_j) + t (2 tau_j - 1): the eq factor of the variable bound in this round
                const Fr w0 = fr_sub(one, tau[j]), dw = fr_sub(fr_add(tau[j], tau[j]), one), w2 = fr_add(w0, fr_add(dw, dw)), w3 = fr_add(w2, dw);
                e[0] = fr_mul(fr_mul(cj, w0), e[0]); e[1] = fr_mul(fr_mul(cj, w2), e[1]); e[2] = fr_mul(fr_mul(cj, w3), e[2]);
            } else host_cubic_evals(tail1, e);
            tw += now_ms() - ta;
            Fr ev[4] = {e[0], fr_sub(st.claim, e[0]), e[1], e[2]};
            ta = now_ms();
            RoundPart1 p1 = sumcheck_round_begin(P.sc1, j, ev, 4, st, g, g.sc_4, tr);
            tb += now_ms() - ta; ta = now_ms();
            P.rx[j] = p1.r_j;
            if (j < ndev) {
                if (armed(j + 1)) c.go(&p1.r_j, 1); else fold_launch(j + 1, &p1.r_j);
                if (armed(j + 2)) fold_launch(j + 2, nullptr);
                // eq(tau_j, r_j) = tau_j r_j + (1 - tau_j)(1 - r_j)
                cj = fr_mul(cj, fr_add(fr_mul(tau[j], p1.r_j), fr_mul(fr_sub(one, tau[j]), fr_sub(one, p1.r_j))));
            }
            tl += now_ms() - ta; ta = now_ms();
            sumcheck_round_finish(P.sc1, j, p1, st, g, g.sc_4, tr);             // overlaps the device fold
            tf += now_ms() - ta;
            if (j >= ndev && !tail_built) build_tail();
            if (j >= ndev) for (auto &t : tail1) host_fold_top(t, p1.r_j);
        }
        if (getenv("OTTI_TRACE")) fprintf(stderr, "[otti] phase1 rounds=%zu (device %zu) wait %.3f begin %.3f launch %.3f finish %.3f ms\n", nrx, ndev + 1, tw, tb, tl, tf);
        blind_claim_postsc1 = st.blinds_evals[nrx - 1];
        for (int k = 0; k < 4; k++) c.h_results[8 + k] = tail1[k][0];
    }
    // what phase two needs from rx alone — eq(rx, .) (the full table on every rank: a column needs every row) and the working copy of
    // z — is queued now, so the device builds it while the host runs the sigma protocols between the phases
    dev_eq_evals(c, P.rx.data(), nrx, S.T[0].p, S.eqs.p);
    if (sh) dev_gather_strided(c, wit.z.p, G, rk, S.zw.p, V2l);
    else OTTI_HIP(hipMemcpyAsync(S.zw.p, wit.z.p, 2 * V * sizeof(Fr), hipMemcpyDeviceToDevice, c.stream));
    const Fr tau_claim = c.h_results[8], Az_claim = c.h_results[9], Bz_claim = c.h_results[10], Cz_claim = c.h_results[11];
    T.ms[2] = now_ms() - t0;
    if (hooks && hooks->on_rx) hooks->on_rx(P.rx);

    // ---- claims about Az, Bz, Cz at rx (nizk/mod.rs sigma protocols; host)
    Fr Az_blind = tape.random_scalar("Az_blind"), Bz_blind = tape.random_scalar("Bz_blind"), Cz_blind = tape.random_scalar("Cz_blind"),
       prod_blind = tape.random_scalar("prod_Az_Bz_blind");
    P.pok = knowledge_prove(P.claims_phase2[2], g, tr, tape, Cz_claim, Cz_blind);
    Fr prod = fr_mul(Az_claim, Bz_claim);
    P.prod = product_prove(P.claims_phase2[0], P.claims_phase2[1], P.claims_phase2[3], g, tr, tape, Az_claim, Az_blind, Bz_claim, Bz_blind, prod, prod_blind);
    tr.append_point("comm_Az_claim", P.claims_phase2[0].b); tr.append_point("comm_Bz_claim", P.claims_phase2[1].b);
    tr.append_point("comm_Cz_claim", P.claims_phase2[2].b); tr.append_point("comm_prod_Az_Bz_claims", P.claims_phase2[3].b);
    {
        Fr blind_expected = fr_mul(tau_claim, fr_sub(prod_blind, Cz_blind));
        Fr claim_post = fr_mul(fr_sub(prod, Cz_claim), tau_claim);
        P.eq1 = equality_prove(g, tr, tape, claim_post, blind_expected, claim_post, blind_claim_postsc1);
    }

    // ---- phase two set-up: r_A,r_B,r_C, eq(rx), ABC = r_A*A(rx,.) + r_B*B(rx,.) + r_C*C(rx,.)  (K2, K6)
    Fr rA = tr.challenge_scalar("challenege_Az"), rB = tr.challenge_scalar("challenege_Bz"), rC = tr.challenge_scalar("challenege_Cz");
    Fr claim2 = fr_add(fr_add(fr_mul(rA, Az_claim), fr_mul(rB, Bz_claim)), fr_mul(rC, Cz_claim));
    Fr blind_claim2 = fr_add(fr_add(fr_mul(rA, Az_blind), fr_mul(rB, Bz_blind)), fr_mul(rC, Cz_blind));
    t0 = now_ms();
    {
        Fr coef[3] = {rA, rB, rC};
        dev_spmv3(c, cols_set, S.T[0].p, S.ABC.p, nullptr, nullptr, true, coef);
        c.sync();
    }
    T.ms[3] = now_ms() - t0;

    // ---- sum-check phase two (K7 + K4): comb = z * ABC
    t0 = now_ms();
    P.ry.resize(nry);
    Fr claims_phase2[2], blind_claim_postsc2;
    {
        SumcheckState st;
        sumcheck_draw_tape(st, tape, nry, 3);
        st.pre = std::move(early2.pre);
        st.claim = claim2; st.blind_claim = blind_claim2;
        { Term t2[2] = {{g.sc_1.G[0], claim2}, {g.sc_1.h, blind_claim2}}; g.commit_terms_c(st.comm_claim.b, t2, 2); }
        P.sc2.comm_polys.resize(nry); P.sc2.comm_evals.resize(nry); P.sc2.proofs.resize(nry);
        const size_t s_loc2 = nry - lgG, lgT2 = std::min(s_loc2, kHostTailBits > lgG ? kHostTailBits - lgG : (size_t)0), T2 = (size_t)1 << lgT2, ndev = s_loc2 - lgT2;
        const size_t dsum = T2 >= 2 ? ndev + 1 : ndev;
        std::vector<Fr> tail2[2]; bool tail_built = false;
        auto fetch_tail = [&] { dev_fetch(c, S.zw.p, kTailSlot, T2); dev_fetch(c, S.ABC.p, kTailSlot + (int)T2, T2); OTTI_HIP(hipEventRecord(c.ev0, c.stream)); };
        auto build_tail = [&] {
            OTTI_HIP(hipEventSynchronize(c.ev0));
            const Fr *mine = &c.h_results[kTailSlot];
            std::vector<Fr> all;
            if (sh) { all.resize(2 * T2 * G); sh->allgather(mine, 2 * T2 * sizeof(Fr), all.data()); }
            for (int k = 0; k < 2; k++) {
                tail2[k].resize(T2 * G);
                for (size_t r = 0; r < G; r++) for (size_t i = 0; i < T2; i++) tail2[k][i * G + r] = (sh ? all.data() + 2 * T2 * r : mine)[k * T2 + i];
            }
            tail_built = true;
        };
        const bool arm_ok = c.armed_ok() && T2 >= 2;
        auto armed = [&](size_t k) { return arm_ok && k >= 1 && k <= ndev && (V2l >> (k - 1)) <= kArmMaxLen; };
        std::vector<unsigned long long> tk(ndev + 2, 0);
        auto fold_launch = [&](size_t k, const Fr *r) {                     // as in phase one
            const size_t len = V2l >> (k - 1);
            if (len >= 4) tk[k] = r ? dev_sc_quad_fold_eval(c, S.zw.p, S.ABC.p, len, *r, 0) : dev_sc_quad_fold_eval_armed(c, S.zw.p, S.ABC.p, len, 0);
            else { dev_fold_top(c, S.zw.p, len, *r); dev_fold_top(c, S.ABC.p, len, *r); }
            if (k == ndev) fetch_tail();
        };
        tk[0] = dev_sc_quad_eval(c, S.zw.p, S.ABC.p, V2l, 0);
        if (ndev == 0) fetch_tail(); else if (armed(1)) fold_launch(1, nullptr);
        bool idle_told = false;
        for (size_t j = 0; j < nry; j++) {
            Fr e[2];
            if (j >= dsum && !tail_built) build_tail();
            if (j < dsum) {
                c.wait_ticket(tk[j]);
                e[0] = c.h_results[0]; e[1] = c.h_results[1];
                if (sh) sh->allreduce_fr_device(c.results.p, nullptr, e, 2);
            } else host_quad_evals(tail2, e);
            Fr ev[3] = {e[0], fr_sub(st.claim, e[0]), e[1]};
            RoundPart1 p1 = sumcheck_round_begin(P.sc2, j, ev, 3, st, g, g.sc_3, tr);
            P.ry[j] = p1.r_j;
            if (j < ndev) {
                if (armed(j + 1)) c.go(&p1.r_j, 1); else fold_launch(j + 1, &p1.r_j);
                if (armed(j + 2)) fold_launch(j + 2, nullptr);
            }
            if (hooks && hooks->on_idle && !idle_told && (j >= ndev || (V2l >> (j + 1)) <= kArmMaxLen)) { idle_told = true; hooks->on_idle(); }   // the bandwidth-bound rounds are behind us
            sumcheck_round_finish(P.sc2, j, p1, st, g, g.sc_3, tr);
            if (j >= ndev && !tail_built) build_tail();
            if (j >= ndev) for (auto &t : tail2) host_fold_top(t, p1.r_j);
        }
        blind_claim_postsc2 = st.blinds_evals[nry - 1];
        claims_phase2[0] = tail2[0][0]; claims_phase2[1] = tail2[1][0];
    }
    T.ms[4] = now_ms() - t0;
    if (hooks && hooks->on_ry) hooks->on_ry(P.ry);

    // ---- polyeval: poly_vars.evaluate(ry[1..]) + PolyEvalProof::prove (K9, K10)
    t0 = now_ms();
    Fr blind_eval;
    {
        const Fr *r = P.ry.data() + 1; const size_t lv = ell / 2;
        std::vector<Fr> Lv_host = eq_evals_host(r, lv);                             // L-side table also needed on the host for LZ_blind
        dev_eq_evals2(c, r, lv, S.Lv.p, r + lv, ell - lv, S.Rv.p, S.eqs.p);
        dev_poly_bound(c, my_rows, Ll, Rsz, S.Lv.p + rk * Ll, S.LZ.p, S.bound_scratch.p);
        if (sh) {                                                                    // partial L^T Z of this rank's rows -> sum over ranks
            std::vector<Fr> lz(Rsz);
            OTTI_HIP(hipMemcpyAsync(lz.data(), S.LZ.p, Rsz * sizeof(Fr), hipMemcpyDeviceToHost, c.stream));
            c.sync();
            sh->allreduce_fr(lz.data(), Rsz);
            OTTI_HIP(hipMemcpyAsync(S.LZ.p, lz.data(), Rsz * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
            c.sync();                                                                // lz is about to go out of scope
        }
        // Z(ry[1..]) = <L^T Z, R> because eq(r, i) factors as L[i_hi] * R[i_lo]
        dev_dot(c, S.LZ.p, S.Rv.p, Rsz, 12);
        Fr LZ_blind = fr_zero();
        for (size_t i = 0; i < Lsz; i++) LZ_blind = fr_add(LZ_blind, fr_mul(blinds_vars[i], Lv_host[i]));
        blind_eval = tape.random_scalar("blind_eval");
        tr.append_protocol_name("polynomial evaluation proof");
        const PcView pv = {g.pc_n.h, g.pc_1.G[0], g.pc_1.h, Rsz};
        const PeBufs pb = {S.LZ.p, S.Rv.p, S.a.p, S.s.p, S.b2.p, S.s2.p, S.rows.p, S.extras.p};
        P.polyeval = dplog_prove_device(c, DG, g, pv, pb, LZ_blind, nullptr, blind_eval, P.comm_vars_at_ry, tr, tape);
    }
    T.ms[5] = now_ms() - t0;

    // ---- final equality: z(ry) * ABC(ry) against the phase-two claim
    {
        Fr blind_eval_Z = fr_mul(fr_sub(fr_one(), P.ry[0]), blind_eval);
        Fr blind_expected = fr_mul(claims_phase2[1], blind_eval_Z);
        Fr claim_post = fr_mul(claims_phase2[0], claims_phase2[1]);
        P.eq2 = equality_prove(g, tr, tape, claim_post, blind_expected, claim_post, blind_claim_postsc2);
    }
}

std::vector<uint8_t> nizk_prove_resident(Instance &I, DeviceWitness &wit, Gens &g, const void *tlabel, size_t tlabel_len, const uint8_t *seed32,
                                         ProveTimings *tm, ShardComm *sh) {
    DevCtx &c = DevCtx::get();
    ActiveProof active;
    SpinPool::Session pool_session;                               // helper threads spin for the duration of this proof
    const double t_start = now_ms(); ProveTimings T{};
    Transcript tr(tlabel, tlabel_len);
    RandomTape tape(seed32);
    NizkProof P;
    tr.append_protocol_name("Spartan NIZK proof");
    r1cs_prove_device(I, wit, g, tr, tape, P, T, sh);
    std::vector<uint8_t> out = P.serialize();
    T.ms[6] = now_ms() - t_start;
    OTTI_HIP(hipStreamSynchronize(c.stream));                    // already drained: every result above was waited for
    KStats::get().flush();
    if (tm) *tm = T;
    return out;
}

std::vector<uint8_t> nizk_prove_gpu(Instance &I, const std::vector<Fr> &vars_padded, const std::vector<Fr> &inputs, Gens &g,
                                    const void *tlabel, size_t tlabel_len, const uint8_t *seed32, ProveTimings *tm) {
    DeviceWitness w(I, vars_padded, inputs);
    return nizk_prove_resident(I, w, g, tlabel, tlabel_len, seed32, tm);
}

}  // namespace otti
