// gfx950 (MI355X, CDNA4) kernels of the Spartan NIZK proving path (this file: the device context, kernel timing, host waits;
// kernels live in k_field.hip, k_sparse.hip, k_sumcheck.hip, k_msm.hip).  Wave64; 256 CUs in 8 XCDs; every kernel here is integer
// work on 256-bit field elements (8 x u32 limbs, v_mad_u64_u32 chains) — no MFMA applies.  Streaming kernels (K1-K7, K9)
// move 32-byte elements with two 16-byte accesses per lane and are HBM-bound; the fixed-base MSM (K8) is integer-ALU-bound.
//
// Kernel <-> upstream hot loop [RECALL: the reference's Spartan/ submodule is empty]:
//   k_spmv3_*            K1 sparse_mlpoly.rs SparseMatPolynomial::multiply_vec, K6 compute_eval_table_sparse (transposed copy)
//   k_eq_tree/_expand    K2 dense_mlpoly.rs EqPolynomial::evals
//   k_sc_*               K3/K7 sumcheck.rs prove_cubic_with_additive_term / prove_quad inner loops, fused with
//                        K4 dense_mlpoly.rs DensePolynomial::bound_poly_var_top of the previous round
//   k_fold_top/_bot      K4/K5 bound_poly_var_top / bound_poly_var_bot
//   k_msm_rows/_finish   K8 DensePolynomial::commit_inner -> Commitments::commit (dalek vartime_multiscalar_mul), also K10's L/R
//   k_poly_bound_*       K9 DensePolynomial::bound
//   k_bullet_step        K10 nizk/bullet.rs BulletReductionProof::prove scalar bookkeeping
#include "kernels_common.h"
#include "snark_dev.h"
#include <atomic>

namespace otti {

void hip_check(hipError_t e, const char *what, const char *file, int line) {
    if (e == hipSuccess) return;
    char buf[512]; snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); throw OutOfDeviceMemory(OTTI_ERR_NO_DEVICE, buf); }
    throw Error(OTTI_ERR_NO_DEVICE, buf);
}

// ------------------------------------------------------------------------------------------------ context
// One context per calling thread: its own stream, mailbox, result buffers.  Provers running on different threads of one process
// therefore overlap on the device (one proof's latency-bound rounds leave most of the chip idle) while sharing the read-only HBM
// objects — instance CSR, generator window table, resident witness.
// Contexts live in a process-wide pool and are never destroyed: a thread leases one on its first call and hands it back when it
// exits (no HIP call may run from a thread_local destructor — the runtime, or a profiler hooked into it, may already be gone).
namespace {
struct CtxPool { std::mutex mu; std::vector<DevCtx *> idle; bool probed = false, failed = false; std::string why; int dev = 0, num_cu = 0; };
CtxPool &ctx_pool() { static CtxPool *p = new CtxPool(); return *p; }
struct CtxLease {
    DevCtx *c = nullptr;
    ~CtxLease() { if (c) { CtxPool &P = ctx_pool(); std::lock_guard<std::mutex> lk(P.mu); P.idle.push_back(c); } }
};
}  // namespace
static void mail_alloc(DevCtx &c, void **p, size_t bytes) {
    if (hipHostMalloc(p, bytes, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) return;
    (void)hipGetLastError(); c.host_coherent = false;
    OTTI_HIP(hipHostMalloc(p, bytes, hipHostMallocDefault));
}
DevCtx &DevCtx::get() {
    thread_local CtxLease lease;
    if (lease.c) return *lease.c;
    CtxPool &P = ctx_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (P.failed) throw Error(OTTI_ERR_NO_DEVICE, P.why);
        if (!P.probed) {
            try {
                int count = 0;
                hipError_t e = hipGetDeviceCount(&count);
                if (e != hipSuccess || count == 0) throw Error(OTTI_ERR_NO_DEVICE, "no HIP device visible: the MI355X proving path has no CPU fallback");
                const char *env = getenv("OTTI_DEVICE"); if (!env) env = getenv("LOCAL_RANK");
                if (env) P.dev = atoi(env) % count;
                OTTI_HIP(hipSetDevice(P.dev));
                hipDeviceProp_t prop; OTTI_HIP(hipGetDeviceProperties(&prop, P.dev));
                if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
                    char buf[256]; snprintf(buf, sizeof buf, "device %d is %s; this library carries gfx950 code objects only", P.dev, prop.gcnArchName);
                    throw Error(OTTI_ERR_NO_DEVICE, buf);
                }
                P.num_cu = prop.multiProcessorCount; P.probed = true;
            } catch (const Error &e) { P.failed = true; P.why = e.what(); throw; }
        }
    }
    OTTI_HIP(hipSetDevice(P.dev));                             // the current device is a per-thread setting
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.idle.empty()) { lease.c = P.idle.back(); P.idle.pop_back(); return *lease.c; }
    }
    std::unique_ptr<DevCtx> c(new DevCtx());
    c->device = P.dev; c->num_cu = P.num_cu;
    OTTI_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->partials.alloc((size_t)kMaxBlocks * 4);
    c->results.alloc(kResultSlots);
    // Words a kernel spins on (h_go) or mails results and flags to (h_results, h_flag) must be fine-grained coherent host memory: asked
    // for explicitly (HIP_HOST_COHERENT=0 in a caller's environment would otherwise make the default allocation non-coherent and an
    // armed kernel would never see go()).  If the runtime refuses the flags, the default allocation is used and launches are never armed.
    c->host_coherent = true;
    auto host_alloc = [&](void **p, size_t bytes) { mail_alloc(*c, p, bytes); };
    host_alloc((void **)&c->h_results, kResultSlots * sizeof(Fr));
    OTTI_HIP(hipHostGetDevicePointer((void **)&c->d_results_alias, c->h_results, 0));
    memset(c->h_results, 0, 4 * sizeof(Fr));
    c->h_flag = reinterpret_cast<unsigned long long *>(&c->h_results[3]);         // slots 0..2 + the flag: one line (device.h, kLineMark)
    c->d_flag_alias = reinterpret_cast<unsigned long long *>(&c->d_results_alias[3]);
    c->d_counter.alloc(1);
    OTTI_HIP(hipMemset(c->d_counter.p, 0, sizeof(unsigned)));
    c->d_counts.alloc(2);
    host_alloc((void **)&c->h_go, sizeof(GoBox));
    memset(c->h_go, 0, sizeof(GoBox));
    OTTI_HIP(hipHostGetDevicePointer((void **)&c->d_go_alias, c->h_go, 0));
    c->d_go.alloc((kGoCopies * kGoCopyStride + sizeof(GoBox) - 1) / sizeof(GoBox) + 1);       // kGoCopies copies, kGoCopyStride bytes apart (device.h)
    OTTI_HIP(hipMemset(c->d_go.p, 0, c->d_go.n * sizeof(GoBox)));
    OTTI_HIP(hipEventCreate(&c->ev0)); OTTI_HIP(hipEventCreate(&c->ev1));
    // first launch from this library: makes the runtime load and register its code object NOW (tens of ms for a module of this size) —
    // a one-shot process creates its context on a thread of its own while the input is being parsed
    dev_fill_one(*c, c->results.p, 1);
    OTTI_HIP(hipStreamSynchronize(c->stream));
    lease.c = c.release();
    return *lease.c;
}

void DevCtx::ensure_points(size_t rows, size_t splits) {
    // Growing these buffers invalidates results of launches still in flight (h_points in particular is read by the host after an event
    // wait), so capacities start generous and the prover sizes them for the whole proof before its first launch; a later growth
    // drains the stream first.
    const size_t want_partial = std::max<size_t>(rows * splits, 8192), want_rows = std::max<size_t>(rows, 2 * kHostPtsCap);
    if (want_partial > msm_partial_cap) { if (stream) OTTI_HIP(hipStreamSynchronize(stream)); msm_partial.alloc(want_partial); msm_partial_cap = want_partial; }
    if (want_rows > points_cap) {
        if (stream) OTTI_HIP(hipStreamSynchronize(stream));
        if (h_points) (void)hipHostFree(h_points);
        OTTI_HIP(hipHostMalloc((void **)&h_points, want_rows * 32, hipHostMallocDefault));
        if (hipHostGetDevicePointer((void **)&d_points_host, h_points, 0) != hipSuccess) { (void)hipGetLastError(); d_points_host = nullptr; }
        d_points.alloc(want_rows * 32); msm_final.alloc(want_rows); points_cap = want_rows;
    }
    if (!h_pts) {
        mail_alloc(*this, (void **)&h_pts, kHostPtsCap * sizeof(Pt));
        OTTI_HIP(hipHostGetDevicePointer((void **)&d_pts_alias, h_pts, 0));
        d_counter2.alloc(1); OTTI_HIP(hipMemset(d_counter2.p, 0, sizeof(unsigned)));
    }
}

// ------------------------------------------------------------------------------------------------ kernel timing
KStats &KStats::get() { thread_local KStats s; return s; }            // per calling thread, like the context it times
int KStats::begin(DevCtx &c, int k) {
    if (!on || !((mask >> k) & 1u)) return -1;
    if (pool.empty()) { pool.resize(16384); for (auto &e : pool) OTTI_HIP(hipEventCreate(&e)); cls.resize(8192); }
    if (used + 1 > cls.size()) return -1;                     // pool exhausted until the next flush
    int rec = (int)used++;
    cls[rec] = k;
    OTTI_HIP(hipEventRecord(pool[2 * rec], c.stream));
    return rec;
}
void KStats::end(DevCtx &c, int rec) { if (rec >= 0) (void)hipEventRecord(pool[2 * rec + 1], c.stream); }
void KStats::flush() {
    for (size_t r = 0; r < used; r++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pool[2 * r], pool[2 * r + 1]) == hipSuccess) { total_ms[cls[r]] += ms; count[cls[r]]++; }
    }
    used = 0;
}
void KStats::reset() { used = 0; for (int k = 0; k < KC_COUNT; k++) { total_ms[k] = 0; count[k] = 0; } }

// A second stream for bandwidth-bound work queued beside a chain of latency-bound rounds (SNARK mode, opt-in OTTI_HASH_AHEAD=1: the hash
// layer's pass beside the memory circuits' sum-check).  ONE per process, made on first use and kept, given back at exit like the bulk stream.
// OTTI_SIDE_CUS=32..: confined to the FIRST that many CUs (hipExtStreamCreateWithCUMask); measured at 32 / 64 / 96 against none, the rounds
// beside it cost the same (what they lose is not CU time: profiles/r4_hash_layer_ab.txt), so the default is an ordinary stream.
static hipStream_t side_masked_stream() {
    static std::once_flag once; static hipStream_t ss = nullptr;
    std::call_once(once, [] {
        const char *e = getenv("OTTI_SIDE_CUS");
        const int cus = e ? atoi(e) : 0, ncu = DevCtx::get().num_cu, words = (ncu + 31) / 32;
        if (cus >= 32 && cus < ncu && words <= 8) {
            uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = (i < cus / 32) ? 0xffffffffu : 0u;
            if (hipExtStreamCreateWithCUMask(&ss, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); ss = nullptr; }
        }
        if (!ss && hipStreamCreateWithFlags(&ss, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ss = nullptr; }
        static hipStream_t to_destroy; to_destroy = ss;
        if (ss) atexit([] { if (to_destroy) { (void)hipStreamSynchronize(to_destroy); (void)hipStreamDestroy(to_destroy); to_destroy = nullptr; } });
    });
    return ss;
}
hipStream_t DevCtx::side_stream() {
    if (!ev_side) OTTI_HIP(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
    if (!side) side = side_masked_stream();
    return side;
}
Mailbox DevCtx::next_mailbox(int slot) {
    Mailbox mb; mb.partials = partials.p; mb.counter = d_counter.p; mb.host_results = d_results_alias; mb.host_flag = d_flag_alias;
    static const bool line_env = [] { const char *e = getenv("OTTI_LINE_MAIL"); return !(e && e[0] == '0'); }();
    mb.seq = ++seq; mb.slot = slot; mb.dev_results = results.p; mb.line_mail = (line_env && slot == 0 && host_coherent) ? 1 : 0; return mb;
}
static std::atomic<int> g_active_proofs{0};
ActiveProof::ActiveProof() { g_active_proofs.fetch_add(1, std::memory_order_relaxed); }
int ActiveProof::count() { return g_active_proofs.load(std::memory_order_relaxed); }
ActiveProof::~ActiveProof() { g_active_proofs.fetch_sub(1, std::memory_order_relaxed); }
// kernel classes that have armed launches: timing one of them with HIP events would time the host's part of the round too
constexpr unsigned kArmedClasses = (1u << KC_MSM_SMALL) | (1u << KC_SC_CUBIC) | (1u << KC_SC_QUAD) | (1u << KC_PC_ROUND);
bool DevCtx::armed_ok() const {
    static const bool env_on = [] { const char *e = getenv("OTTI_ARMED"); return !(e && e[0] == '0'); }();
    const KStats &ks = KStats::get();
    return env_on && host_coherent && !(ks.on && (ks.mask & kArmedClasses)) && g_active_proofs.load(std::memory_order_relaxed) <= 1;
}
// the per-round sums of a sharded proof for the RCCL transport: packed into u64 lanes where they were produced (no host pack, no upload)
__global__ void k_fr_to_lanes(const Fr *src, Fr factor, int scale, unsigned long long *lanes, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = src[i]; if (scale) v = fr_mul(v, factor);
    for (int k = 0; k < 8; k++) lanes[8 * i + k] = v.v[k];
}
void dev_fr_to_lanes(hipStream_t st, const Fr *d_src, const Fr *factor, unsigned long long *d_lanes, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_to_lanes, (unsigned)((n + 63) / 64), 64, 0, st, d_src, factor ? *factor : fr_zero(), factor ? 1 : 0, d_lanes, n);
}
// ONE stream for the process confined to all but a few CUs (hipExtStreamCreateWithCUMask), for chip-filling fixed-base MSM launches that run
// BESIDE latency-bound rounds: SNARK mode's ahead-of-time derefs rows (snark_prover.cpp), and the witness commitments while several
// proofs are in flight (prover.cpp) — an MSM workgroup holds its CU's registers for the whole launch, and without the mask a round's
// kernel on another stream waits for one to end (profiles/r3_cumask_probe.txt: 2 ms against 18 us; profiles/r4_inflight_*: rounds
// of 52-84 us on average with six proofs in flight).  Made on first use and kept: creating a stream while a proof runs stalls launches.
// OTTI_DEREFS_FREE_CUS: CUs left out of the mask (32 ... 192, default 64); OTTI_DEREFS_CUMASK=0: no masked stream (callers fall back).
hipStream_t bulk_masked_stream() {
    static std::once_flag once; static hipStream_t ms = nullptr;
    std::call_once(once, [] {
        const char *e = getenv("OTTI_DEREFS_CUMASK"), *f = getenv("OTTI_DEREFS_FREE_CUS");
        const int free_cus = 32 * (f ? std::max(1, std::min(6, atoi(f) / 32)) : 2);
        // the mask is cut to the device's own CU count (a partition has fewer than 256); with fewer than 128 CUs there is nothing worth setting aside
        const int ncu = DevCtx::get().num_cu, free_words = free_cus / 32, words = (ncu + 31) / 32;
        if ((e && e[0] == '0') || ncu < 128 || words > 8 || free_words >= words) { ms = nullptr; return; }
        uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = (i < free_words || i >= words) ? 0u : (i == words - 1 && (ncu & 31)) ? ((1u << (ncu & 31)) - 1u) : 0xffffffffu;
        if (hipExtStreamCreateWithCUMask(&ms, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); ms = nullptr; }
        // given back before the runtime's own exit handlers run (this one is registered after them): a process that still held a CU-masked stream
        // at exit faulted in its teardown under rocprofv3
        static hipStream_t to_destroy; to_destroy = ms;
        if (ms) atexit([] { if (to_destroy) { (void)hipStreamSynchronize(to_destroy); (void)hipStreamDestroy(to_destroy); to_destroy = nullptr; } });
    });
    return ms;
}
static int go_pollers() { static const int m = [] { const char *e = getenv("OTTI_GO_POLLERS"); int v = e ? atoi(e) : 0; return (v >= 1 && v <= 8) ? v : 1; }(); return m; }
static int relay_mode() { static const int m = [] { const char *e = getenv("OTTI_RELAY"); return (e && e[0] == '0') ? 0 : 1; }(); return m; }
Armed DevCtx::arm() { Armed a; a.host = d_go_alias; a.dev = d_go.p; a.want = ++go_issued; a.deadline = arm_deadline; a.relay = relay_mode(); a.pollers = go_pollers(); return a; }
Armed DevCtx::arm_many(int count) { Armed a; a.host = d_go_alias; a.dev = d_go.p; a.want = go_issued + 1; a.deadline = arm_deadline; a.relay = relay_mode(); a.pollers = go_pollers(); go_issued += (unsigned long long)count; return a; }
void DevCtx::go(const Fr *v, int n) {
    if (go_published >= go_issued) throw Error(OTTI_ERR_INTERNAL, "go() without an armed launch");
    if (n > 4) n = 4;
    for (int i = 0; i < n; i++) h_go->v[i] = v[i];
    h_go->tag = go_tag(go_published + 1, v, n);
    __atomic_store_n(&h_go->seq, ++go_published, __ATOMIC_RELEASE);
}
void DevCtx::go_abort() {
    if (go_published >= go_issued && !__atomic_load_n(&h_go->timed_out, __ATOMIC_ACQUIRE)) return;
    __atomic_store_n(&h_go->seq, ~0ull, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(stream);
    go_published = go_issued;
    __atomic_store_n(&h_go->timed_out, 0ull, __ATOMIC_RELEASE);
    __atomic_store_n(&h_go->seq, go_published, __ATOMIC_RELEASE);
    reset_arrival_counters();                                // the context goes back to the pool clean
}
// An armed grid that was released by an abort or a deadline returns before its arrival count is complete, and a launch that was cut
// short for any other reason may have counted in part: the next launch on this context must not inherit that.  Stream idle.
void DevCtx::reset_arrival_counters() {
    (void)hipMemsetAsync(d_counter.p, 0, sizeof(unsigned), stream);
    if (d_counter2.p) (void)hipMemsetAsync(d_counter2.p, 0, sizeof(unsigned), stream);
    (void)hipStreamSynchronize(stream);
}
void DevCtx::wait_ticket(unsigned long long ticket) {
    volatile unsigned long long *f = h_flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        if (*f >= ticket) {
            // a line mail (device.h kLineMark): number and tag came in one 16-byte store, the sums' first half possibly not yet — read until the tag fits
            if (*f == ticket && ((f[1] ^ ticket ^ kLineMark) & 0xffffffffull) == 0) {
                for (unsigned tries = 0;; tries++) {
                    Fr s3[3]; const unsigned long long tag = __atomic_load_n(&h_flag[1], __ATOMIC_ACQUIRE);
                    for (int k = 0; k < 3; k++) s3[k] = h_results[k];
                    if (line_tag(ticket, s3) == tag || *f != ticket) break;
                    if (tries > 50000000u) throw Error(OTTI_ERR_INTERNAL, "a round's mailed line never became whole");
#if defined(__x86_64__)
                    _mm_pause();
#endif
                }
            }
            return;
        }
#if defined(__x86_64__)
        _mm_pause();
#endif
        if ((spins & 0x3ff) == 0x3ff && __atomic_load_n(&h_go->timed_out, __ATOMIC_ACQUIRE)) {
            // an armed launch gave up waiting for this thread (it was stopped for longer than the launch's deadline): its grid returned
            // without touching anything, the launches queued behind it are released the same way, and the proof fails cleanly
            go_abort();
            throw Error(OTTI_ERR_INTERNAL, "an armed launch gave up waiting for the host (the proving thread was stalled beyond the launch's deadline)");
        }
        if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            if (go_published < go_issued) go_abort();         // release whatever is still armed, drain, clear the arrival counters
            else { OTTI_HIP(hipStreamSynchronize(stream)); if (*f >= ticket) return; reset_arrival_counters(); }   // surfaces a device fault as an error instead of spinning forever
            if (*f >= ticket) return;
            throw Error(OTTI_ERR_INTERNAL, "sum-check round result never arrived");
        }
    }
}
void DevCtx::ensure_tail_mail() {
    if (h_tail) return;
    mail_alloc(*this, (void **)&h_tail, (size_t)kTailMaxGroups * sizeof(TailMail));
    memset(h_tail, 0, (size_t)kTailMaxGroups * sizeof(TailMail));
    OTTI_HIP(hipHostGetDevicePointer((void **)&d_tail_alias, h_tail, 0));
}
static long tail_timeout_ms() { static const long v = [] { const char *e = getenv("OTTI_TAIL_TIMEOUT_MS"); long x = e ? atol(e) : 0; return x > 0 ? x : 5000L; }(); return v; }
void DevCtx::wait_tail(int n_groups, unsigned long long want) {
    const auto t0 = std::chrono::steady_clock::now();
    int done = 0;                                             // lines [0, done) have arrived
    for (unsigned spins = 0;; spins++) {
        while (done < n_groups && __atomic_load_n(&h_tail[done].seq, __ATOMIC_ACQUIRE) >= want) done++;
        if (done == n_groups) return;
#if defined(__x86_64__)
        _mm_pause();
#endif
        if ((spins & 0x3ff) == 0x3ff && __atomic_load_n(&h_go->timed_out, __ATOMIC_ACQUIRE)) {
            go_abort();
            throw Error(OTTI_ERR_INTERNAL, "the persistent sum-check launch gave up waiting for the host (the proving thread was stalled beyond the launch's deadline)");
        }
        // a round of the persistent launch takes tens of microseconds; seconds without every line in mean that part of its grid is not
        // resident (the workgroups that are wait for the host, the host for all of them): give the launch up — the caller proves again without it
        if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(tail_timeout_ms())) {
            if (go_published < go_issued) go_abort(); else { (void)hipStreamSynchronize(stream); reset_arrival_counters(); }
            throw TailTimeout(OTTI_ERR_INTERNAL, "sum-check round result never arrived (persistent launch: its grid was not resident as a whole)");
        }
    }
}
// the same wait, adding up the W partial sums of every instance as its lines come in (lines a few ahead are prefetched: each is a fresh
// cache line the device has just written, and 144 dependent misses in a row would cost more than the round's arithmetic)
// lines [i0, i1) of a round's mails: wait for each (bounded when `bounded`: a helper thread must not throw) and add the W partials of every instance up
static bool tail_sum_range(TailMail *h_tail, int i0, int i1, int W, unsigned long long want, Fr *sums, bool bounded) {
#if defined(__x86_64__)
    for (int i = i0; i < i1 && i < i0 + 16; i++) { _mm_prefetch((const char *)&h_tail[i], _MM_HINT_T0); _mm_prefetch((const char *)&h_tail[i] + 64, _MM_HINT_T0); }
#endif
    for (int i = i0; i < i1; i++) {
#if defined(__x86_64__)
        if (i + 16 < i1) { _mm_prefetch((const char *)&h_tail[i + 16], _MM_HINT_T0); _mm_prefetch((const char *)&h_tail[i + 16] + 64, _MM_HINT_T0); }
#endif
        if (bounded) {
            unsigned spins = 0;
            while (__atomic_load_n(&h_tail[i].seq, __ATOMIC_ACQUIRE) < want) {
                if (++spins > 2000000u) return false;             // ~ a millisecond or more: the calling thread takes the slow path with its failure handling
#if defined(__x86_64__)
                _mm_pause();
#endif
            }
        }
        // the line came in one store instruction but as two 64-byte halves nothing orders: read, check the tag, read again until it fits (normally at once)
        Fr part[3];
        for (unsigned tries = 0;; tries++) {
            const unsigned long long s = __atomic_load_n(&h_tail[i].seq, __ATOMIC_ACQUIRE), tag = __atomic_load_n(&h_tail[i].tag, __ATOMIC_ACQUIRE);
            for (int k = 0; k < 3; k++) part[k] = h_tail[i].s[k];
            if (s == want && go_tag(s, part, 3) == tag) break;
            if (tries > 4000000u) { if (bounded) return false; throw Error(OTTI_ERR_INTERNAL, "a sum-check round's mail never became whole"); }
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
        Fr *acc = sums + 3 * (i / W);
        if (i % W == 0) { acc[0] = part[0]; acc[1] = part[1]; acc[2] = part[2]; }
        else { acc[0] = fr_add(acc[0], part[0]); acc[1] = fr_add(acc[1], part[1]); acc[2] = fr_add(acc[2], part[2]); }
    }
    return true;
}
void DevCtx::wait_tail_sums(int n_inst, int W, unsigned long long want, Fr *sums /* [n_inst][3] */) {
    const int n = n_inst * W;
    // 128-144 lines of 3 partial sums: adding them up on one core cost 2.6 us of every round (profiles/r3_tail_stamps.txt); the instances are
    // independent, so the prover thread's helpers (pool.h: pinned next to it, ~55 ns hand-over) take a share each
    SpinPool &pool = SpinPool::get();
    const int nt = (n >= 48 && n_inst >= 2) ? std::min(std::min(4, pool.workers() + 1), n_inst) : 1;
    if (nt > 1) {
        bool ok[4] = {true, true, true, true};
        std::function<void()> tasks[4];
        for (int t = 0; t < nt; t++) {
            const int y0 = n_inst * t / nt, y1 = n_inst * (t + 1) / nt;
            tasks[t] = [this, t, y0, y1, W, want, sums, &ok] { ok[t] = tail_sum_range(h_tail, y0 * W, y1 * W, W, want, sums, true); };
        }
        pool.parallel(tasks, nt);
        if (ok[0] && ok[1] && ok[2] && ok[3]) return;
    }
    wait_tail(n, want);                                       // (throws if the launch gave up or never answers)
    tail_sum_range(h_tail, 0, n, W, want, sums, false);
}
void DevCtx::wait_points(unsigned long long ticket) {
    if (!ticket) { sync(); return; }
    wait_ticket(ticket);
    encode_pending();
}
void DevCtx::sync() {
    OTTI_HIP(hipStreamSynchronize(stream));
    encode_pending();
}
void DevCtx::encode_pending() {
    if (pending_host_encode >= 2) {
        SpinPool &pool = SpinPool::get(); const int nt = std::min<int>(pool.workers() + 1, (int)pending_host_encode);
        const size_t n = pending_host_encode;
        std::vector<std::function<void()>> tasks(nt);
        for (int t = 0; t < nt; t++) tasks[t] = [this, t, nt, n] { for (size_t i = t; i < n; i += nt) pt_encode(h_points + 32 * i, h_pts[i]); };
        pool.parallel(tasks.data(), nt);
    } else if (pending_host_encode == 1) pt_encode(h_points, h_pts[0]);
    pending_host_encode = 0;
}

}  // namespace otti
