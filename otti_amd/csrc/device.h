// Device layer of the proving path: HBM-resident objects and the kernel launch functions (k_*.hip).
// Everything runs on one HIP stream owned by DevCtx; results that the Fiat-Shamir transcript needs come back through
// a small pinned buffer.  No function here falls back to the host: without a gfx950 device they throw Error(OTTI_ERR_NO_DEVICE).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <functional>
#include <cstddef>
#include "spartan.h"

namespace otti {

#define OTTI_HIP(expr) ::otti::hip_check((expr), #expr, __FILE__, __LINE__)
void hip_check(hipError_t e, const char *what, const char *file, int line);
struct OutOfDeviceMemory : Error { using Error::Error; };
// the persistent sum-check tail (snark_dev.h) never answered: its grid was not resident as a whole (another tenant of the GPU, a CU mask, a
// partition).  SNARK::prove catches this one, switches the tail off for the process and proves again with a launch per round.
struct TailTimeout : Error { using Error::Error; };   // hipErrorOutOfMemory: callers that can make do with less catch this one

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    DevBuf() {}
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void alloc(size_t count) { release(); if (count) OTTI_HIP(hipMalloc((void **)&p, count * sizeof(T))); n = count; }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// three CSR matrices side by side (A, B, C), by value into kernels
struct DCsr3 { const uint32_t *ptr[3]; const uint32_t *idx[3]; const Fr *val[3]; const int32_t *small[3]; };
struct DeviceCsrSet {
    DevBuf<uint32_t> ptr[3], idx[3]; DevBuf<Fr> val[3];
    DevBuf<int32_t> small[3];                                 // per entry: the coefficient as a small signed integer, or kNotSmall (then val[] is read)
    bool use_small = false;                                   // most coefficients are small integers (a compiled circuit): the kernels read 4 bytes per entry instead of 32
    DevBuf<uint32_t> heavy;                                   // ids of rows whose longest list exceeds kHeavyRow
    DevBuf<uint32_t> seg_row, seg_no, seg_begin;              // their segments (row id, segment number), and each long row's first segment
    size_t rows = 0, n_heavy = 0, n_seg = 0;
    double avg_row = 0.0;                                     // entries per row and matrix: picks the row-per-lane or the row-per-quad kernel
    DCsr3 view() const { DCsr3 v; for (int k = 0; k < 3; k++) { v.ptr[k] = ptr[k].p; v.idx[k] = idx[k].p; v.val[k] = val[k].p; v.small[k] = small[k].p; } return v; }
};
struct DeviceInstance { DeviceCsrSet by_row, by_col; size_t nnz = 0; };
// Rank k of g holds the constraint rows r = k (mod g) (renumbered r / g, columns untouched: multiply_vec output lands where the
// low-bit-sharded phase-one tables need it) and, as a second copy, the entries of columns c = k (mod g) (renumbered c / g, rows
// untouched: compute_eval_table_sparse output lands where the phase-two tables need it).  SURVEY.md 8(e).
struct DeviceShard { int rank = 0, world = 1; DeviceCsrSet by_row, by_col; };

// fixed-base window table: entry (base b, window w, digit d in 1..E) = d * 2^(c*w) * P[b] in affine Niels form
struct DeviceGens {
    DevBuf<TabEntry> table; int c = 0, W = 0; size_t E = 0, nbases = 0;
    double build_ms[2] = {0, 0};                            // what building it took: allocations / upload + kernels
    const TabEntry *entry0(size_t base) const { return table.p + base * (size_t)W * E; }
};

// where a sum-check kernel's last workgroup delivers the round's totals (see finish_in_kernel)
struct Mailbox { Fr *partials; unsigned *counter; Fr *host_results; unsigned long long *host_flag; unsigned long long seq; int slot; int line_mail = 0;
                 Fr *dev_results; };   // dev_results: the totals once more in HBM (same slots), for a device-side collective over them (shard.h, RCCL transport)

// Armed launches.  The sequential rounds cost a launch + dispatch (10-15 us) on top of the kernel itself when the kernel can only be
// launched once the host knows the round's challenge.  An ARMED kernel is queued before that: its first workgroup spins on a pinned host
// word until the host publishes the value (DevCtx::go), copies it to HBM for the other workgroups, and the round starts within a PCIe
// read of the challenge being known.  Every spin has a deadline (kArmDeadlineTicks of s_memrealtime: 30 s) and an abort value, so a grid always drains.
// ONE decision per armed launch: only the launch's first workgroup watches the host word and its deadline, and what it decides (the
// value arrived / aborted / gave up) is what every other workgroup acts on (they watch dev->seq alone, with a backstop of
// the leader's deadline plus a quarter: the leader never ran) — a grid never folds in part.  A leader that gives up says so in host->timed_out, so the host fails the
// proof at once instead of waiting for a result that will not come.
// Layout: the sequence number, a tag and the first value share one 64-byte line, so a poll that finds the number it waits for has the
// value in the same batch of loads (one PCIe round trip, not two).  The tag = go_tag(seq, values) makes a batch self-validating: the
// loads of one poll may be served at different times, and a batch that mixes old and new words fails the tag and is simply polled
// again.  timed_out (written by the DEVICE) sits in a line of its own.
struct alignas(64) GoBox { unsigned long long seq, tag, pad[2]; Fr v[4]; unsigned long long timed_out, pad2[3]; };
static_assert(sizeof(GoBox) == 192 && offsetof(GoBox, v) == 32 && offsetof(GoBox, timed_out) == 160, "GoBox layout is read by hand-written loads");
// the same function on both sides of the bus: every 64-bit word of the n values, rotated by its position, folded into the sequence number
HD unsigned long long go_tag(unsigned long long seq, const Fr *v, int n) {
    unsigned long long h = seq * 0x9e3779b97f4a7c15ull;
    for (int k = 0; k < n; k++)
        for (int j = 0; j < 4; j++) {
            const unsigned long long w = (unsigned long long)v[k].v[2 * j] | ((unsigned long long)v[k].v[2 * j + 1] << 32);
            const int rot = ((4 * k + j) * 5 + 1) & 63;
            h ^= (w << rot) | (w >> ((64 - rot) & 63));
        }
    return h;
}
// The leader's copy for the other workgroups exists kGoCopies times, kGoCopyStride bytes apart (different memory channels): with every waiting
// workgroup polling ONE line, 128 pollers kept a single HBM channel busy enough to double the latency of each poll.  Workgroup b polls copy b mod kGoCopies.
constexpr int kGoCopies = 8;
constexpr size_t kGoCopyStride = 4096;
// The completion flag lives in result slot 3 — with slots 0..2 one 128-byte line: {Fr s[3]; u64 seq; u64 tag; u64 pad[2]}.  A launch whose K <= 3
// totals go to slot 0 mails that line with ONE store instruction and no fence (Mailbox.line_mail; the persistent tail does the same, snark_dev.h):
// number and tag travel in one 16-byte store; the low 32 bits of the tag are the number's xor kLineMark, which tells the host that the line's
// first half is covered by the tag (a fenced mail leaves an older tag behind, whose number does not fit) and must be checked.
constexpr unsigned long long kLineMark = 0x5a5a5a5aull;
HD unsigned long long line_tag(unsigned long long seq, const Fr *s3) { return (go_tag(seq, s3, 3) & ~0xffffffffull) | ((seq ^ kLineMark) & 0xffffffffull); }
struct Armed { GoBox *host; GoBox *dev; unsigned long long want, deadline; int relay = 1, pollers = 1; };   // relay 0: the copy's number is polled alone and the values loaded after it (OTTI_RELAY=0; A/B)   // want == 0: not armed (values come as kernel arguments); deadline in 100 MHz ticks
constexpr unsigned long long kArmDeadlineTicks = 3000000000ull;   // 30 s of s_memrealtime: longer than any host stall the prover's own 20 s result wait tolerates
constexpr size_t kArmMaxLen = 65536;                         // sum-check tables up to this length fold in <= 64 workgroups: only those launches are armed

struct DevCtx {
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr; hipEvent_t ev_side = nullptr;   // a second stream for work that runs beside a latency-bound stage (SNARK mode: hash-layer evaluations beside the second batched sum-check); made on first use
    hipStream_t side_stream();
    int device = 0, num_cu = 256;
    DevBuf<Fr> partials;                                      // [kMaxBlocks][4] per-block partial sums
    DevBuf<Fr> spmv_partial;                                  // 3 partial sums per long-row segment of the SpMV in flight
    DevBuf<Fr> results;                                       // small device result slots
    Fr *h_results = nullptr;                                  // pinned mirror of `results`
    Fr *d_results_alias = nullptr;                            // device address of h_results (zero-copy stores)
    unsigned long long *h_flag = nullptr, *d_flag_alias = nullptr, seq = 0;   // host-visible completion flag of the latest mailbox launch
    DevBuf<unsigned> d_counter;                               // arrival counter of the publishing workgroups
    DevBuf<unsigned long long> d_counts;                      // two 64-bit tallies for the kernels that count (non-canonical / small scalars): no allocation per call
    Mailbox next_mailbox(int slot);
    GoBox *h_go = nullptr, *d_go_alias = nullptr; DevBuf<GoBox> d_go; unsigned long long go_issued = 0, go_published = 0;
    unsigned long long arm_deadline = kArmDeadlineTicks;     // of the launches armed from now on (the self-test shortens it)
    bool host_coherent = false;                               // the words kernels spin on / mail to are fine-grained coherent host memory (else: no armed launches)
    void reset_arrival_counters();                            // after an aborted or timed-out launch: a grid may have left them non-zero (stream must be idle)
    hipEvent_t ev_order = nullptr;                            // orders a caller's stream (otti_kd_*) against this context's own
    std::vector<struct RowSumSlot *> row_slots;               // the verifier's variable-base sums in flight on this context (prover.cpp), buffers kept across proofs
    struct TailMail *h_tail = nullptr, *d_tail_alias = nullptr;   // per-workgroup mail lines of the persistent sum-check tail (snark_dev.h), pinned
    void ensure_tail_mail();
    void wait_tail(int n_groups, unsigned long long seq);     // spin until every line carries seq (same failure handling as wait_ticket)
    void wait_tail_sums(int n_inst, int W, unsigned long long seq, Fr *sums);   // the same for a round's mails, summing the W partials of every instance as they arrive
    Armed arm_many(int count);                                // reserves `count` consecutive go() numbers for one persistent launch; .want = the first
    bool armed_ok() const;                                    // off under OTTI_ARMED=0, while kernel classes are being timed (a waiting kernel's duration includes the host), and
                                                              // while another proof is in flight in this process (a waiting grid holds wave slots the other proof's kernels could use: measured -15 % throughput with six in flight)
    Armed arm();                                              // for the next launch; the k-th armed launch consumes the k-th go()
    void go(const Fr *v, int n);                              // publish up to four values to the oldest armed launch that has none yet
    void go_abort();                                          // release every armed launch still waiting (they exit without touching their data) and drain the stream
    void wait_ticket(unsigned long long ticket);              // spin until the launch with that sequence number has delivered
    DevBuf<Pt> msm_keep;                                      // row sums parked on the device (MSM_KEEP)
    DevBuf<Pt> msm_partial, msm_final;                        // [rows][chunks] partial sums, [rows] row sums
    Pt *h_pts = nullptr; size_t pending_host_encode = 0;      // pinned: row sums of small launches, compressed on the host in sync()
    uint8_t *h_points = nullptr;                              // pinned: compressed points coming back
    uint8_t *d_points_host = nullptr;                         // the device's address of h_points: the encode kernel writes there as well (no copy engine round trip)
    DevBuf<uint8_t> d_points;
    size_t msm_partial_cap = 0, points_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    static DevCtx &get();                                     // the calling thread's context; throws Error(OTTI_ERR_NO_DEVICE) when no device is usable
    DevCtx() {} DevCtx(const DevCtx &) = delete; DevCtx &operator=(const DevCtx &) = delete;
    struct Scratch *scratch = nullptr;                        // the prover's HBM workspace (prover.cpp); travels with the context
    struct SnarkScratch *snark_scratch = nullptr;             // SNARK mode's per-proof buffers (snark_prover.cpp), kept across proofs like the above
    void sync();
    void wait_points(unsigned long long ticket);              // results of a dev_msm_rows launch: flag wait when fused, else stream sync
    void encode_pending();
    Pt *d_pts_alias = nullptr; DevBuf<unsigned> d_counter2;
    void ensure_points(size_t rows, size_t splits);
};
struct RowSumSlot { DevBuf<uint8_t> comp; DevBuf<Niels> pts; DevBuf<Fr> sc; DevBuf<Pt> out; DevBuf<unsigned> bad; bool busy = false; };
hipStream_t bulk_masked_stream();                                    // the process's CU-masked stream for chip-filling MSM launches beside latency-bound rounds (k_context.hip); nullptr if there is none
struct ActiveProof { ActiveProof(); ~ActiveProof(); static int count(); };       // RAII around one prove call: counts the proofs in flight in this process
constexpr int kResultSlots = 8192;                          // 256 KB pinned: round sums, sum-check tails (SNARK: up to 18 x 3 tables x 128 elements)
constexpr size_t kHostEncodeRows = 8;
constexpr size_t kHostPtsCap = 512;

// per-kernel-class HIP-event timing on the library's own stream (bench.py's roofline numbers come from here)
enum KClass { KC_MSM_ROWS = 0, KC_MSM_SMALL, KC_MSM_FINISH, KC_SC_CUBIC, KC_SC_QUAD, KC_SPMV, KC_EQ, KC_REDUCE, KC_BOUND, KC_BULLET, KC_OTHER,
              KC_PC_ROUND, KC_PROD_LAYER, KC_HASH_LAYER, KC_GATHER, KC_DOT_MANY /* SNARK mode (k_snark.hip) */,
              KC_DECODE, KC_MSM_VAR /* verifier (k_msm.hip) */, KC_COUNT };
struct KStats {
    bool on = false; unsigned mask = 0xffffffffu;            // bit k set: kernel class k is timed
    std::vector<hipEvent_t> pool; std::vector<int> cls; size_t used = 0;
    double total_ms[KC_COUNT] = {0}; unsigned long long count[KC_COUNT] = {0};
    static KStats &get();
    int begin(DevCtx &c, int k);                              // returns record index or -1
    void end(DevCtx &c, int rec);
    void flush();                                             // stream must be idle
    void reset();
};
struct KScope { DevCtx &c; int rec; KScope(DevCtx &c_, int k) : c(c_), rec(KStats::get().begin(c_, k)) {} ~KScope() { KStats::get().end(c, rec); } };

// witness resident in HBM: z = vars || 1 || inputs || 0..  (2 * num_vars Montgomery-form elements)
struct DeviceWitness {
    DevBuf<Fr> z; std::vector<Fr> inputs;
    double small_fraction = 0.0;                              // share of the variables below 2^128: picks the MSM variant of the commitment
    DeviceWitness(const Instance &I, const std::vector<Fr> &vars_padded, const std::vector<Fr> &inputs);
    // straight from the caller's canonical bytes: validation (InvalidScalar) and the conversion to Montgomery form happen on the device
    DeviceWitness(const Instance &I, const uint8_t *vars32, size_t nvars, const std::vector<Fr> &inputs);
};
void ensure_device_objects(Instance &I, Gens &g);          // lazily built, shared by every prover thread (guarded)
void ensure_instance_device(Instance &I);
void ensure_gens_device(Gens &g);
void release_gens_device(Gens &g);                                     // frees the window table; the next ensure_gens_device rebuilds it
int device_window_bits(size_t nbases);                   // window width of the fixed-base table (prover.cpp)
// R1CSInstance::evaluate on the device: (A,B,C)(rx,ry) = <eq(rx), M * eq(ry)> for M in {A,B,C}  (verifier's O(nnz + N + V) work)
constexpr int kInstEvalSlot = 16;                            // result slots of the instance evaluation (nothing a verifier launches in between writes them)
void instance_evaluate_begin(Instance &I, const std::vector<Fr> &rx, const std::vector<Fr> &ry);      // queued on the calling thread's stream
void instance_evaluate_finish(Fr out[3]);                                                              // same thread: waits, reads
void instance_evaluate_gpu(Instance &I, const std::vector<Fr> &rx, const std::vector<Fr> &ry, Fr out[3]);
// sh == nullptr (or a world of one): the whole proof on this GPU.  Otherwise this process proves its shard of the SAME proof as the
// other ranks of sh (collective call: same instance, witness, generators, label and seed on every rank); every rank returns the
// same bytes, equal to the single-GPU proof.
class ShardComm;
std::vector<uint8_t> nizk_prove_resident(Instance &I, DeviceWitness &wit, Gens &g, const void *tlabel, size_t tlabel_len, const uint8_t *seed32,
                                         ProveTimings *tm, ShardComm *sh = nullptr);
// hooks: called on the proving thread the moment the first (rx) / second (ry) sum-check's challenges are complete — SNARK mode starts the
// work that depends on them alone (the two halves of the dereferenced polynomial and their commitment rows) on another stream while the
// R1CS proof goes on through its latency-bound rounds
struct R1csHooks { std::function<void(const std::vector<Fr> &)> on_rx, on_ry; std::function<void()> on_idle; };   // on_idle: from here on the proof is short rounds only (the chip is idle between them)
void r1cs_prove_device(Instance &I, DeviceWitness &wit, Gens &g, Transcript &tr, RandomTape &tape, NizkProof &P, ProveTimings &T, ShardComm *sh, const R1csHooks *hooks = nullptr);
std::shared_ptr<DeviceShard> upload_instance_shard(const Instance &I, int rank, int world);
// DotProductProofLog::prove on the device (prover.cpp): generator stream indices of (gens_n.h, gens_1.G[0], gens_1.h) and the row length
struct PeBufs { Fr *LZ, *Rv, *a, *s, *b2, *s2, *rows, *extras; };      // R elements each (rows: 2 R, extras: 4 (log2 R + 1))
DotProductProofLog dplog_prove_device(DevCtx &c, const DeviceGens &DG, const Gens &g, const PcView &v, const PeBufs &B, const Fr &LZ_blind, const Fr *y_known,
                                      const Fr &blind_y, CPoint &Cy_out, Transcript &tr, RandomTape &tape);
size_t dev_witness_ingest(DevCtx &c, Fr *z, size_t n, size_t *n_small = nullptr);   // returns the number of non-canonical scalars (zeroed); n_small: how many are below 2^128
void dev_gather_strided(DevCtx &c, const Fr *in, size_t stride, size_t offset, Fr *out, size_t n);   // out[i] = in[i*stride + offset]

std::shared_ptr<DeviceInstance> upload_instance(const Instance &I);
std::shared_ptr<DeviceGens> build_device_gens(const Gens &g, int window_bits);

// ---- element-wise / conversion
void dev_fr_op(DevCtx &c, int op, const Fr *a, const Fr *b, Fr *out, size_t n);
void dev_from_canonical(DevCtx &c, const Fr *in, Fr *out, size_t n);      // raw LE integer -> Montgomery (must be < l)
void dev_to_canonical(DevCtx &c, const Fr *in, Fr *out, size_t n);
void dev_fill_zero(DevCtx &c, Fr *p, size_t n);
// ---- K1 / K6: sparse products.  combine == false: out[k][r] = sum_p val_k[p] * x[idx_k[p]];  true: out[0][r] = sum_k coef[k] * (...)
void dev_spmv3(DevCtx &c, const DeviceCsrSet &m, const Fr *x, Fr *out0, Fr *out1, Fr *out2, bool combine, const Fr coef[3]);
// ---- K2: eq tables.  r is a HOST array (challenges come from the transcript)
void dev_eq_evals(DevCtx &c, const Fr *r_host, size_t ell, Fr *out, Fr *scratch /* >= 3 * 4096 elements; 5 * 4096 for ell = 25 */);
void dev_eq_evals2(DevCtx &c, const Fr *r0_host, size_t ell0, Fr *out0, const Fr *r1_host, size_t ell1, Fr *out1, Fr *scratch);   // both in one launch when each has at most 13 variables
// ---- K3/K4/K5/K7: sum-check rounds.  Results land in c.h_results[slot .. slot+k)
// each returns a ticket: c.wait_ticket(ticket) returns once h_results[slot..] hold that launch's sums (no stream synchronise)
unsigned long long dev_sc_cubic_eval(DevCtx &c, const Fr *A, const Fr *B, const Fr *C, const Fr *D, size_t len, int slot);
unsigned long long dev_sc_cubic_fold_eval(DevCtx &c, Fr *A, Fr *B, Fr *C, Fr *D, size_t len, const Fr &r, int slot);   // len >= 4; folds to len/2, sums over the folded tables
// phase one with the eq table factored out (see k_sumcheck.hip): E[i] = hi ? hi[i >> lo_bits] * lo[i & (2^lo_bits - 1)] : lo[i]
struct EqSrc { const Fr *hi, *lo; int lo_bits;
               uint32_t stride = 1, offset = 0;      // table index of a kernel's item i is i * stride + offset (a rank's residue class of a sharded table; 1, 0 otherwise)
               // one more variable ABOVE the tabulated ones (eq_at only): E[i] = (bit top_bit of i ? top : 1 - top) * table[i mod 2^top_bit].  The product
               // circuits' pyramids leave a layer's first variable out — it is the last challenge to be drawn, and without it they are built ahead of time
               int top_bit = -1; Fr top; };
void dev_eq_pyramid(DevCtx &c, const Fr *r_host, size_t n, Fr *out /* 2^(n+1) - 1 elements: level k at out + 2^k - 1 */);
void dev_eq_pyramid2(DevCtx &c, const Fr *r0_host, size_t n0, Fr *out0, const Fr *r1_host, size_t n1, Fr *out1);   // two in one launch (out1 may be null)
unsigned long long dev_sc_cubic3_eval(DevCtx &c, const Fr *B, const Fr *C, const Fr *D, size_t len, const EqSrc &E, int slot);
unsigned long long dev_sc_cubic3_fold_eval(DevCtx &c, Fr *B, Fr *C, Fr *D, size_t len, const Fr &r, const EqSrc &E, int slot);
// armed variants (see Armed above): the fold challenge is the next value published with c.go()
unsigned long long dev_sc_cubic3_fold_eval_armed(DevCtx &c, Fr *B, Fr *C, Fr *D, size_t len, const EqSrc &E, int slot);
unsigned long long dev_sc_quad_fold_eval_armed(DevCtx &c, Fr *A, Fr *B, size_t len, int slot);
unsigned long long dev_sc_quad_eval(DevCtx &c, const Fr *A, const Fr *B, size_t len, int slot);
unsigned long long dev_sc_quad_fold_eval(DevCtx &c, Fr *A, Fr *B, size_t len, const Fr &r, int slot);
void dev_fold_top(DevCtx &c, Fr *Z, size_t len, const Fr &r);
void dev_fold_bot(DevCtx &c, const Fr *Z, Fr *out, size_t len, const Fr &r);
void dev_fetch(DevCtx &c, const Fr *src, int slot, size_t n);              // async copy of n elements into h_results[slot..]
// ---- K8: fixed-base MSM rows.  Row i: sum_j dense[i*stride + j] * P[j] (j < n_dense) + sum_e extra_s[i*n_extra+e] * P[extra_base[e]]
// Compressed results land in c.h_points[32*i ..] after c.sync(); they also stay in c.d_points.
enum { MSM_COMPRESSED = 0, MSM_RAW = 1, MSM_KEEP = 2 };
// returns a ticket: c.wait_points(ticket) returns once the compressed points are in c.h_points (ticket 0 = plain stream sync)
unsigned long long dev_msm_rows(DevCtx &c, const DeviceGens &g, const Fr *dense, size_t stride, size_t n_dense, size_t rows, const Fr *extra_s,
                                const uint32_t *extra_base_host, size_t n_extra, int mode = MSM_COMPRESSED, const Pt *addend = nullptr,
                                bool sparse_hint = false);
// sparse_hint: the dense scalars are mostly small numbers (DeviceWitness::small_fraction) — bulk launches then compact the non-zero
// (term, window) pairs into a work list instead of giving every pair a lane.  Results are identical either way.
double dev_small_fraction(DevCtx &c, const Fr *z, size_t n);               // share of scalars below 2^128 (synchronises the stream)
// MSM_RAW: skip compression; after c.sync() the extended row sums are in c.h_pts[0..rows).
// MSM_KEEP: no output; the row sums stay on the device in c.msm_keep (to be passed as `addend` of a later launch, which then
// compresses (row sum + addend)).  Lets the host draw the blinds while the device already sums the witness terms.
// ---- K9: LZ[j] = sum_i Lv[i] * Z[i*R + j]
void dev_poly_bound(DevCtx &c, const Fr *Z, size_t L, size_t R, const Fr *Lv, Fr *out, Fr *scratch /* >= 64*R */);
// chunk sums of the same bound over eq(rest) alone (k_sumcheck.hip): out = (L / m) x R; false (nothing launched) when the geometry does not allow it
bool dev_poly_bound_chunks(DevCtx &c, const Fr *Z, size_t L, size_t R, const Fr *Lv_rest, size_t m, Fr *out, Fr *scratch /* >= 64*R */);
// dot product of two device vectors -> h_results[slot]
void dev_dot(DevCtx &c, const Fr *a, const Fr *b, size_t n, int slot);
// ---- K10: bullet reduction bookkeeping on the ORIGINAL generators (see prover.cpp)
// One launch per round.  If fold_first, a and b (length 2*n_cur) are folded to n_cur with (u, u_inv) and s is updated; then, for
// n_cur >= 2, the next round's c_L, c_R go to extra_out[0], extra_out[2] and the dense scalar rows sL, sR (length R each) to rows[0..2R).
void dev_bullet_step(DevCtx &c, Fr *a, Fr *b, Fr *s, size_t R, size_t n_cur, bool fold_first, const Fr &u, const Fr &u_inv, Fr *rows, Fr *extra_out);
void dev_bullet_finish(DevCtx &c, Fr *a, Fr *b, Fr *s, size_t R, const Fr &u, const Fr &u_inv, const Fr &d, Fr *rows, int slot_a);   // last fold (2 -> 1), a, b to result slots slot_a, slot_a + 1, rows = d s
// One bullet-reduction round as ONE launch: applies the previous challenge (fold), derives the scalars of L and R from the round state
// and sums both rows (fused finish; compressed L, R arrive in c.h_points[0..64) after c.wait_points(ticket)).  extra_s: 4 scalars
// {unused, blind_L, unused, blind_R} (the c_L / c_R terms are computed in the kernel); extra_base: {Q, H}.
// armed (device.h): u and u_inv are not known yet; the launch takes {u, u_inv, raw(u), raw(u_inv)} from the next c.go()
unsigned long long dev_bullet_round(DevCtx &c, const DeviceGens &g, size_t R, size_t n_cur, bool fold, const Fr &u, const Fr &u_inv, const Fr *a_in,
                                    const Fr *b_in, const Fr *s_in, Fr *a_out, Fr *b_out, Fr *s_out, const Fr *extra_s, const uint32_t *extra_base,
                                    bool armed = false);
// ---- verifier: decompression of n ristretto255 points into affine Niels form (bad: count of encodings that do not decode), and the
// variable-base MSM over them: out[w * splits + s] = sum over the points of split s of digit_w(scalar) * point — window sums the host
// combines (253 doublings: a sequential chain a host core runs 30 x faster than a GPU lane).  Returns the window width c it used.
void dev_decode_niels(DevCtx &c, const uint8_t *compressed_dev, size_t n, Niels *out, unsigned *bad);
int dev_msm_var(DevCtx &c, const Niels *pts, const Fr *scalars, size_t n, Pt *out, size_t out_cap, int *n_windows, int *n_splits);
double dev_madd_peak(DevCtx &c);                                          // mixed point additions per second, whole chip (the MSM's ALU roof)
// v[i] (times *factor when given: a device-resident scalar is not needed, it travels by value) as 8 u32 limbs widened to u64 lanes, on `st`
void dev_fr_to_lanes(hipStream_t st, const Fr *d_src, const Fr *factor, unsigned long long *d_lanes, size_t n);
double dev_fr_mul_peak(DevCtx &c);                                        // Montgomery products in GF(l) per second, whole chip (the streaming kernels' second roof)
void dev_scale(DevCtx &c, const Fr *in, const Fr &k, Fr *out, size_t n);
void dev_fill_one(DevCtx &c, Fr *p, size_t n);

}  // namespace otti
