// extern "C" surface of libottispartan (include/otti_spartan.h).  No exception crosses this boundary.
#include "device.h"
#include "shard.h"
#include "snark.h"
#include "hosttail.h"
#include <array>
#include <atomic>
#include <chrono>
#include "pool.h"
#include <mutex>

using namespace otti;

struct otti_instance { std::unique_ptr<Instance> I; };
struct otti_gens { std::unique_ptr<Gens> g; };
struct otti_witness { std::unique_ptr<DeviceWitness> w; };
struct otti_snark_gens { std::unique_ptr<SnarkGens> g; };
struct otti_comp_comm { std::unique_ptr<CompComm> c; };

otti_r1cs *otti_r1cs_from(size_t nc, size_t nv, size_t ni, const std::vector<otti_entry> &A, const std::vector<otti_entry> &B,
                          const std::vector<otti_entry> &C, const std::vector<uint8_t> &vars, const std::vector<uint8_t> &inputs);
otti_r1cs *zkif_load_impl(const char *circuit_path, const char *inputs_path, const char *witness_path);
void zkif_write_impl(const otti_r1cs *r, const char *circuit_path, const char *inputs_path, const char *witness_path);

static thread_local std::string g_last_error;
template <class F> static int32_t guarded(F &&f) {
    try { g_last_error.clear(); return f(); }
    catch (const Error &e) { g_last_error = e.what(); return e.code; }
    catch (const std::bad_alloc &) { g_last_error = "out of memory"; return OTTI_ERR_INTERNAL; }
    catch (const std::exception &e) { g_last_error = e.what(); return OTTI_ERR_INTERNAL; }
    catch (...) { g_last_error = "unknown error"; return OTTI_ERR_INTERNAL; }
}
static std::vector<Fr> scalars_from_bytes(const uint8_t *b, size_t n) {
    std::vector<Fr> v(n);
    for (size_t i = 0; i < n; i++) if (!fr_from_bytes(v[i], b + 32 * i)) throw Error(OTTI_ERR_INVALID_SCALAR, "non-canonical scalar in assignment");
    return v;
}
// VarsAssignment::new + pad, InputsAssignment::new
static void load_assignment(const Instance &I, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs, std::vector<Fr> &vars, std::vector<Fr> &inputs) {
    if (nvars > I.num_vars) throw Error(OTTI_ERR_INVALID_NUM_VARS, "more variables than the instance has");
    if (ninputs != I.num_inputs) throw Error(OTTI_ERR_INVALID_NUM_INPUTS, "wrong number of inputs");
    vars = scalars_from_bytes(vars32, nvars); vars.resize(I.num_vars, fr_zero());
    inputs = scalars_from_bytes(inputs32, ninputs);
}

extern "C" {

size_t otti_last_error(char *buf, size_t cap) {
    if (buf && cap) { size_t n = std::min(cap - 1, g_last_error.size()); memcpy(buf, g_last_error.data(), n); buf[n] = 0; }
    return g_last_error.size();
}
void otti_buf_free(void *p) { free(p); }

int32_t otti_device_count(void) {
    // asked once: on a host without a GPU every hipGetDeviceCount call re-probes the driver (~0.2 s of system time)
    static const int count = [] { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }();
    return count;
}

int32_t otti_host_selftest(uint32_t iterations) {
    return guarded([&] {
        auto g = gens_new(16, 16, 1);
        Shake256 xof; xof.absorb("otti-host-selftest", 18);
        for (uint32_t it = 0; it < iterations; it++) {
            uint8_t w[64]; xof.squeeze(w, 64);
            Fr s = fr_from_bytes_wide(w);
            if (it == 0) s = fr_zero(); if (it == 1) s = fr_one(); if (it == 2) s = fr_neg(fr_one());
            xof.squeeze(w, 64);
            const Pt rnd = pt_from_uniform_bytes(w);
            // five-limb round trip and compression against the generic code
            uint8_t a[32], b[32];
            pt_encode_fast(a, rnd); pt_encode_ref(b, rnd);
            if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "pt_encode_fast differs from pt_encode_ref");
            const Pt back = ptfe_to(ptfe_from(rnd));
            pt_encode_ref(a, back);
            if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "five-limb round trip changed a point");
            // fixed-base table (8-bit windows, five-limb mixed additions) against a variable-base multiplication
            const size_t slot = it % g->small_tables.size();
            size_t base = 0; for (size_t i = 0; i < g->small_slot.size(); i++) if (g->small_slot[i] == (int)slot) base = i;
            Pt acc = rnd; g->small_tables[slot].accumulate(acc, s);
            const Pt want = pt_add(rnd, host_scalarmul(g->P[base], s));
            pt_encode_ref(a, acc); pt_encode_ref(b, want);
            if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "fixed-base table result differs from the variable-base multiplication");
            if (host_ifma_available()) {   // the AVX-512 IFMA mixed addition (hostifma.h) against the scalar five-limb one, both signs, on a point with lazily reduced limbs
                const NielsFe &ne = g->small_tables[slot].t[(it * 37) % g->small_tables[slot].t.size()];
                const Niels4 n4 = niels4_from(ne);
                for (int neg = 0; neg < 2; neg++) {
                    PtFe x1 = ptfe_from(rnd), x2 = x1;
                    for (int k = 0; k < 3; k++) { ptfe_madd(x1, ne, neg != 0); ifma_madd(x2, n4, neg != 0); }
                    pt_encode_fe(a, x1); pt_encode_fe(b, x2);
                    if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "IFMA mixed addition differs from the scalar one");
                }
            }
            {   // host_scalarmul (windowed; AVX-512 IFMA doublings and additions where the CPU has them) against plain double-and-add in the generic
                // 4 x u64 code of point.h, and a small multi-scalar sum against the sum of the single products
                const Fr raw = fr_to_raw(s);
                Pt ref = pt_identity();
                for (int bit = 255; bit >= 0; bit--) { ref = pt_dbl(ref); if ((raw.v[bit / 32] >> (bit % 32)) & 1) ref = pt_add(ref, rnd); }
                pt_encode_ref(a, host_scalarmul(rnd, s)); pt_encode_ref(b, ref);
                if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "host_scalarmul differs from double-and-add");
                const Fr s3[3] = {s, fr_neg(s), fr_add(s, fr_one())}; const Pt p3[3] = {rnd, g->P[1], g->P[2]};
                Pt sum = pt_add(pt_add(host_scalarmul(p3[0], s3[0]), host_scalarmul(p3[1], s3[1])), host_scalarmul(p3[2], s3[2]));
                pt_encode_ref(a, host_msm(s3, p3, 3)); pt_encode_ref(b, sum);
                if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "host_msm differs from the sum of its terms");
            }
            // the host's last sum-check rounds (hosttail.h): the AVX-512 IFMA form against the scalar one on random tables of every size, both kinds of instance
            {   // the division-step inversion against the exponentiation (the fast one falls back to the other if its own check fails: also count that it did not)
                Fr inv_fast; const Fr inv_ref = fr_inv(s);
                if (!fr_inv_fast_try(s, inv_fast)) throw Error(OTTI_ERR_INTERNAL, "fr_inv_fast gave up on an input");
                if (!fr_eq(inv_ref, inv_fast)) throw Error(OTTI_ERR_INTERNAL, "fr_inv_fast differs from fr_inv");
            }
            if (it < 128) hosttail_selftest(it);
            {   // the four-way split multiplication (verifier rounds) against the plain one
                SplitTable st; split_table_build(st, rnd);
                pt_encode_ref(a, split_table_mul(st, s)); pt_encode_ref(b, host_scalarmul(rnd, s));
                if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "split_table_mul differs from host_scalarmul");
            }
            {   // five-limb extended addition against the generic one
                PtFe x = ptfe_from(rnd); ptfe_add(x, ptfe_from(want));
                pt_encode_fe(a, x); pt_encode_ref(b, pt_add(rnd, want));
                if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "five-limb point addition differs from pt_add");
                x = ptfe_identity(); ptfe_add(x, ptfe_from(rnd)); pt_encode_fe(a, x); pt_encode_ref(b, rnd);
                if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "five-limb addition to the identity changed a point");
            }
            {   // five-limb decompression against the generic one (valid encodings, and one that is not)
                Pt d1, d2; pt_encode_ref(a, rnd);
                if (!pt_decode_fast(d1, a) || !pt_decode(d2, a)) throw Error(OTTI_ERR_INTERNAL, "a valid encoding did not decode");
                pt_encode_ref(b, d1); if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "pt_decode_fast does not invert the encoding");
                if (memcmp(&d1, &d2, sizeof(Pt)) && !(fp_eq(d1.X, d2.X) && fp_eq(d1.Y, d2.Y) && fp_eq(d1.T, d2.T))) throw Error(OTTI_ERR_INTERNAL, "pt_decode_fast differs from pt_decode");
                a[0] ^= 1;                                        // negative s: both must refuse
                if (pt_decode_fast(d1, a) != pt_decode(d2, a)) throw Error(OTTI_ERR_INTERNAL, "pt_decode_fast and pt_decode disagree on a non-canonical encoding");
                for (int k = 0; k < 32; k++) a[k] = (uint8_t)(w[k] ^ (it * 37 + k));   // arbitrary bytes: mostly invalid, sometimes valid
                a[31] &= 0x7f;
                const bool ok1 = pt_decode_fast(d1, a), ok2 = pt_decode(d2, a);
                if (ok1 != ok2) throw Error(OTTI_ERR_INTERNAL, "pt_decode_fast and pt_decode disagree on arbitrary bytes");
                if (ok1) { pt_encode_ref(b, d1); uint8_t b2[32]; pt_encode_ref(b2, d2); if (memcmp(b, b2, 32)) throw Error(OTTI_ERR_INTERNAL, "pt_decode_fast and pt_decode decode to different points"); }
            }
            {   // multiplication by a small signed integer (the SpMV kernels' path for compiled circuits) against the Montgomery product
                const int32_t cs[6] = {0, 1, -1, 0x7ffffffe, -0x7ffffffe, (int32_t)(w[5] | (w[6] << 8) | (w[7] << 16) | ((w[8] & 0x3f) << 24)) * ((w[9] & 1) ? -1 : 1)};
                for (int32_t cc : cs) {
                    const Fr cm = cc < 0 ? fr_neg(fr_from_u64((uint64_t)(-(int64_t)cc))) : fr_from_u64((uint64_t)cc);
                    if (!fr_eq(fr_mul_small(s, cc), fr_mul(s, cm))) throw Error(OTTI_ERR_INTERNAL, "fr_mul_small differs from fr_mul");
                    if (fr_small_code(cm) != cc) throw Error(OTTI_ERR_INTERNAL, "fr_small_code does not recover a small integer");
                }
                Fr big = fr_from_u64(0x80000000ull); if (fr_small_code(big) != kNotSmall || fr_small_code(fr_neg(big)) != kNotSmall) throw Error(OTTI_ERR_INTERNAL, "fr_small_code accepts 2^31");
                if (it > 2 && fr_small_code(s) != kNotSmall) throw Error(OTTI_ERR_INTERNAL, "fr_small_code accepts a random field element");
            }
            pt_encode_fast(a, pt_identity()); pt_encode_ref(b, pt_identity());
            if (memcmp(a, b, 32)) throw Error(OTTI_ERR_INTERNAL, "identity encodes differently");
        }
        {   // the transcript's fused message operations (hash.h Strobe128::merlin_append / merlin_challenge) against the separate STROBE
            // operations they stand for: random labels and messages of 0 .. 100 bytes, so that every position of the rate block, the
            // block boundary and the long-message fallback are all crossed many times
            Strobe128 fused("Merlin v1.0"), plain("Merlin v1.0");
            for (uint32_t it = 0; it < 40 * iterations + 2000; it++) {
                uint8_t rnd[8]; xof.squeeze(rnd, 8);
                const size_t L = 1 + rnd[0] % 30, n = rnd[1] % 101; const bool chal = (rnd[2] & 3) == 0;
                char label[32]; uint8_t msg[128], o1[128], o2[128];
                xof.squeeze(label, L); xof.squeeze(msg, n ? n : 1);
                const uint8_t len[4] = {(uint8_t)n, 0, 0, 0};
                if (chal) {
                    fused.merlin_challenge(label, L, o1, n);
                    plain.meta_ad(label, L, false); plain.meta_ad(len, 4, true); plain.prf(o2, n, false);
                    if (memcmp(o1, o2, n)) throw Error(OTTI_ERR_INTERNAL, "fused transcript challenge differs from the separate STROBE operations");
                } else {
                    fused.merlin_append(label, L, msg, n);
                    plain.meta_ad(label, L, false); plain.meta_ad(len, 4, true); plain.ad(msg, n, false);
                }
            }
            uint8_t o1[64], o2[64];
            fused.prf(o1, 64, false); plain.prf(o2, 64, false);
            if (memcmp(o1, o2, 64)) throw Error(OTTI_ERR_INTERNAL, "fused transcript operations left a different state");
        }
        return OTTI_OK;
    });
}

// nanoseconds per operation of the host-side primitives on the sequential path (measurement aid: tools/hostbench.py)
int32_t otti_host_tail_bench(uint32_t np, uint32_t nd, uint64_t T, uint32_t threads, uint32_t reps, double out[2]) {
    return guarded([&] {
        if (!out) throw Error(OTTI_ERR_BAD_ARG, "null out pointer");
        if (np > 12 || nd > 12 || T < 2 || T > 4096 || (T & (T - 1))) throw Error(OTTI_ERR_BAD_ARG, "host tail bench: at most 12 + 12 instances, tables of 2 .. 4096 elements");
        hosttail_bench((int)np, (int)nd, (size_t)T, (int)threads, (int)reps, out);
        return OTTI_OK;
    });
}
int32_t otti_host_microbench(double out[10]) {
    return guarded([&] {
        if (!out) throw Error(OTTI_ERR_BAD_ARG, "null out pointer");
        auto g = gens_new(16, 16, 1);
        auto now = [] { return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        Shake256 xof; xof.absorb("otti-host-microbench", 20);
        uint8_t w[64]; xof.squeeze(w, 64); Fr s = fr_from_bytes_wide(w); xof.squeeze(w, 64); Fr s2 = fr_from_bytes_wide(w);
        xof.squeeze(w, 64); const Pt rnd = pt_from_uniform_bytes(w);
        const int R = 2000; double t0; volatile uint8_t sink = 0;
        { PtFe acc = ptfe_from(rnd); t0 = now(); for (int i = 0; i < R; i++) { g->small_tables[0].accumulate(acc, s); s = fr_add(s, s2); } out[0] = (now() - t0) / R; uint8_t b[32]; pt_encode(b, ptfe_to(acc)); sink ^= b[0]; }
        { uint8_t b[32]; Pt p = rnd; t0 = now(); for (int i = 0; i < R; i++) { pt_encode(b, p); p.X.v[0] ^= b[0] & 1; } out[1] = (now() - t0) / R; sink ^= b[1]; }
        { uint64_t st[25] = {1}; t0 = now(); for (int i = 0; i < 10 * R; i++) keccak_f1600(st); out[2] = (now() - t0) / (10 * R); sink ^= (uint8_t)st[3]; }
        { Transcript tr("bench", 5); uint8_t b[32] = {7}; t0 = now(); for (int i = 0; i < R; i++) { tr.append_point("comm_poly", b); Fr c = tr.challenge_scalar("challenge_nextround"); b[0] ^= (uint8_t)c.v[0]; } out[3] = (now() - t0) / R; sink ^= b[0]; }
        { Fr a = s, b = s2; t0 = now(); for (int i = 0; i < 100 * R; i++) a = fr_mul(a, b); out[4] = (now() - t0) / (100 * R); sink ^= (uint8_t)a.v[0]; }
        { Fr a = s; t0 = now(); for (int i = 0; i < R / 10; i++) a = fr_inv(fr_add(a, s2)); out[5] = (now() - t0) / (R / 10); sink ^= (uint8_t)a.v[0]; }
        {   // hand one empty task to a helper thread and wait for it
            SpinPool::Session session; SpinPool &pool = SpinPool::get();
            out[6] = 0;
            if (pool.workers() > 0) { std::atomic<int> n{0}; std::function<void()> f = [&] { n.fetch_add(1, std::memory_order_relaxed); }; for (int i = 0; i < 100; i++) { pool.submit(0, f); pool.wait(0); }
                t0 = now(); for (int i = 0; i < R; i++) { pool.submit(0, f); pool.wait(0); } out[6] = (now() - t0) / R; }
            // one zero-knowledge sum-check round's host work as the prover runs it (cubic round, 4 coefficients), without a device
            const int rounds = 200;
            Transcript tr("bench", 5); RandomTape tape(w);
            SumcheckState st; sumcheck_draw_tape(st, tape, rounds, 4);
            for (auto &p : st.pre) { Term t = {g->sc_4.h, st.blinds_poly[0]}; p.bp_h = g->commit_terms(&t, 1); p.be_h = p.bp_h; p.rb_h = p.bp_h; p.delta = p.bp_h; p.to_fe(); pt_encode(p.delta_c.b, p.delta); }
            st.claim = s; st.blind_claim = s2; pt_encode(st.comm_claim.b, rnd);
            ZKSumcheckProof pf; pf.comm_polys.resize(rounds); pf.comm_evals.resize(rounds); pf.proofs.resize(rounds);
            double tb = 0, tf = 0;
            for (int j = 0; j < rounds; j++) {
                Fr ev[4] = {s, fr_sub(st.claim, s), s2, fr_mul(s, s2)};
                t0 = now(); RoundPart1 p1 = sumcheck_round_begin(pf, j, ev, 4, st, *g, g->sc_4, tr); tb += now() - t0;
                t0 = now(); sumcheck_round_finish(pf, j, p1, st, *g, g->sc_4, tr); tf += now() - t0;
                s = fr_add(s, p1.r_j);
            }
            out[7] = tb / rounds; out[8] = tf / rounds; out[9] = pool.workers() + 1;
        }
        (void)sink;
        return OTTI_OK;
    });
}

int32_t otti_instance_new(uint64_t nc, uint64_t nv, uint64_t ni, const otti_entry *A, size_t nA, const otti_entry *B, size_t nB,
                          const otti_entry *C, size_t nC, otti_instance **out) {
    return guarded([&] {
        if (!out) throw Error(OTTI_ERR_BAD_ARG, "null out pointer");
        auto h = std::make_unique<otti_instance>();
        h->I = instance_new(nc, nv, ni, A, nA, B, nB, C, nC);
        *out = h.release(); return OTTI_OK;
    });
}
void otti_instance_free(otti_instance *p) { delete p; }
int32_t otti_instance_dims(const otti_instance *inst, uint64_t *nc, uint64_t *nv, uint64_t *ni) {
    if (!inst) return OTTI_ERR_BAD_ARG;
    if (nc) *nc = inst->I->num_cons; if (nv) *nv = inst->I->num_vars; if (ni) *ni = inst->I->num_inputs;
    return OTTI_OK;
}
int32_t otti_instance_is_sat(const otti_instance *inst, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs, int32_t *sat) {
    return guarded([&] {
        if (!inst || !sat) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        std::vector<Fr> vars, inputs; load_assignment(*inst->I, vars32, nvars, inputs32, ninputs, vars, inputs);
        *sat = inst->I->is_sat(vars, inputs) ? 1 : 0; return OTTI_OK;
    });
}

int32_t otti_gens_new(uint64_t nc, uint64_t nv, uint64_t ni, otti_gens **out) {
    return guarded([&] {
        if (!out) throw Error(OTTI_ERR_BAD_ARG, "null out pointer");
        auto h = std::make_unique<otti_gens>(); h->g = gens_new(nc, nv, ni); *out = h.release(); return OTTI_OK;
    });
}
void otti_gens_free(otti_gens *p) { delete p; }
int32_t otti_gens_table_info(const otti_gens *gens, uint32_t *window_bits, uint64_t *table_bytes) {
    return guarded([&] {
        if (!gens) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        const DeviceGens *d = gens->g->dev.get();
        if (window_bits) *window_bits = d ? (uint32_t)d->c : 0;
        if (table_bytes) *table_bytes = d ? (uint64_t)d->table.n * sizeof(TabEntry) : 0;
        return OTTI_OK;
    });
}
int32_t otti_gens_build_ms(const otti_gens *gens, double *alloc_ms, double *kernels_ms) {
    return guarded([&] {
        if (!gens) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        const DeviceGens *d = gens->g->dev.get();
        if (alloc_ms) *alloc_ms = d ? d->build_ms[0] : 0.0;
        if (kernels_ms) *kernels_ms = d ? d->build_ms[1] : 0.0;
        return OTTI_OK;
    });
}
int32_t otti_gens_release_device(otti_gens *gens) {
    return guarded([&] {
        if (!gens) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        release_gens_device(*gens->g);
        return OTTI_OK;
    });
}
int32_t otti_gens_points(const otti_gens *gens, uint8_t *out32, size_t count) {
    return guarded([&] {
        if (!gens || count > gens->g->P.size()) throw Error(OTTI_ERR_BAD_ARG, "count exceeds the generator stream");
        for (size_t i = 0; i < count; i++) pt_encode(out32 + 32 * i, gens->g->P[i]);
        return OTTI_OK;
    });
}

int32_t otti_prepare_device(otti_instance *inst, otti_gens *gens) {
    return guarded([&] {
        if (inst) ensure_instance_device(*inst->I);
        if (inst && gens) ensure_device_objects(*inst->I, *gens->g);
        else if (gens) ensure_gens_device(*gens->g);
        else if (!inst) DevCtx::get();                              // neither: just bring the HIP runtime and this process's device context up
        return OTTI_OK;
    });
}

static uint8_t *to_malloc(const std::vector<uint8_t> &v, size_t *len) {
    uint8_t *p = (uint8_t *)malloc(std::max<size_t>(1, v.size())); memcpy(p, v.data(), v.size()); *len = v.size(); return p;
}

int32_t otti_nizk_prove(otti_instance *inst, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs, otti_gens *gens,
                        const uint8_t *tlabel, size_t tlabel_len, const uint8_t *seed32, uint32_t flags, uint8_t **proof, size_t *proof_len,
                        double *stage_ms) {
    return guarded([&] {
        if (!inst || !gens || !proof || !proof_len) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        if (!(flags & OTTI_FLAG_GPU)) throw Error(OTTI_ERR_BAD_ARG, "OTTI_FLAG_GPU is the only proving backend; there is no CPU path");
        if (ninputs != inst->I->num_inputs) throw Error(OTTI_ERR_INVALID_NUM_INPUTS, "wrong number of inputs");
        std::vector<Fr> inputs = scalars_from_bytes(inputs32, ninputs);
        ProveTimings tm{};
        DeviceWitness w(*inst->I, vars32, nvars, inputs);          // VarsAssignment::new (InvalidScalar) is checked on the device
        std::vector<uint8_t> pf = nizk_prove_resident(*inst->I, w, *gens->g, tlabel, tlabel_len, seed32, &tm);
        if (stage_ms) memcpy(stage_ms, tm.ms, sizeof tm.ms);
        *proof = to_malloc(pf, proof_len); return OTTI_OK;
    });
}
int32_t otti_witness_upload(otti_instance *inst, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs, otti_witness **out) {
    return guarded([&] {
        if (!inst || !out) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        if (ninputs != inst->I->num_inputs) throw Error(OTTI_ERR_INVALID_NUM_INPUTS, "wrong number of inputs");
        std::vector<Fr> inputs = scalars_from_bytes(inputs32, ninputs);
        auto h = std::make_unique<otti_witness>(); h->w = std::make_unique<DeviceWitness>(*inst->I, vars32, nvars, inputs);
        *out = h.release(); return OTTI_OK;
    });
}
void otti_witness_free(otti_witness *w) { delete w; }
int32_t otti_nizk_prove_resident(otti_instance *inst, otti_witness *wit, otti_gens *gens, const uint8_t *tlabel, size_t tlabel_len,
                                 const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms) {
    return guarded([&] {
        if (!inst || !wit || !gens || !proof || !proof_len) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        ProveTimings tm{};
        std::vector<uint8_t> pf = nizk_prove_resident(*inst->I, *wit->w, *gens->g, tlabel, tlabel_len, seed32, &tm);
        if (stage_ms) memcpy(stage_ms, tm.ms, sizeof tm.ms);
        *proof = to_malloc(pf, proof_len); return OTTI_OK;
    });
}
int32_t otti_shard_init(const char *segment_name, uint32_t rank, uint32_t world) {
    return guarded([&] {
        if (!segment_name) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        shard_comm_set(nullptr);
        shard_comm_set(new ShardComm(segment_name, (int)rank, (int)world));
        return OTTI_OK;
    });
}
int32_t otti_shard_info(uint32_t *rank, uint32_t *world, uint32_t *transport) {
    return guarded([&] {
        if (!shard_comm()) throw Error(OTTI_ERR_BAD_ARG, "otti_shard_init has not been called");
        if (rank) *rank = (uint32_t)shard_comm()->rank(); if (world) *world = (uint32_t)shard_comm()->world();
        if (transport) *transport = (uint32_t)shard_comm()->transport();
        return OTTI_OK;
    });
}
int32_t otti_shard_finalize(void) { return guarded([&] { shard_comm_set(nullptr); return OTTI_OK; }); }
int32_t otti_shard_allgather(const void *mine, size_t nbytes, void *out) {
    return guarded([&] {
        if (!shard_comm()) throw Error(OTTI_ERR_BAD_ARG, "otti_shard_init has not been called");
        if (!mine || !out) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        shard_comm()->allgather(mine, nbytes, out); return OTTI_OK;
    });
}
int32_t otti_shard_allreduce(uint8_t *scalars32, size_t count) {
    return guarded([&] {
        if (!shard_comm()) throw Error(OTTI_ERR_BAD_ARG, "otti_shard_init has not been called");
        std::vector<Fr> v = scalars_from_bytes(scalars32, count);
        shard_comm()->allreduce_fr(v.data(), count);
        for (size_t i = 0; i < count; i++) fr_to_bytes(scalars32 + 32 * i, v[i]);
        return OTTI_OK;
    });
}
int32_t otti_nizk_prove_sharded(otti_instance *inst, otti_witness *wit, otti_gens *gens, const uint8_t *tlabel, size_t tlabel_len,
                                const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms) {
    return guarded([&] {
        if (!inst || !wit || !gens || !proof || !proof_len) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        if (!shard_comm()) throw Error(OTTI_ERR_BAD_ARG, "otti_shard_init has not been called");
        if (!seed32) throw Error(OTTI_ERR_BAD_ARG, "a sharded proof needs an explicit random-tape seed (the same on every rank)");
        ProveTimings tm{};
        std::vector<uint8_t> pf = nizk_prove_resident(*inst->I, *wit->w, *gens->g, tlabel, tlabel_len, seed32, &tm, shard_comm());
        if (stage_ms) memcpy(stage_ms, tm.ms, sizeof tm.ms);
        *proof = to_malloc(pf, proof_len); return OTTI_OK;
    });
}
int32_t otti_nizk_verify(const otti_instance *inst, const uint8_t *inputs32, size_t ninputs, const otti_gens *gens, const uint8_t *tlabel,
                         size_t tlabel_len, const uint8_t *proof, size_t proof_len) {
    return guarded([&] {
        if (!inst || !gens || !proof) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        std::vector<Fr> inputs = scalars_from_bytes(inputs32, ninputs);
        // The verifier is host code (as in the reference); only its O(nnz + N + V) step — evaluating A, B, C at (rx, ry) — goes to the
        // device when one is present.  (Verification is not the proving hot path: without a device it simply stays on the host.)
        // The evaluation is QUEUED here (its point is in the proof) and collected where the verifier needs it, after the sum-check rounds.
        bool queued = false;
        if (inst->I->num_cons >= 4096 && otti_device_count() > 0) {
            try {
                NizkProof P = NizkProof::parse(proof, proof_len);
                instance_evaluate_begin(*const_cast<Instance *>(inst->I.get()), P.rx, P.ry); queued = true;
            } catch (const Error &) { queued = false; }
        }
        const InstEvalFetch fetch = [](Fr out[3]) { instance_evaluate_finish(out); };
        int rc = nizk_verify(*inst->I, inputs, *gens->g, tlabel, tlabel_len, proof, proof_len, nullptr, queued ? &fetch : nullptr);
        if (rc) g_last_error = "proof rejected";
        return rc;
    });
}

// ------------------------------------------------------------------------------------------------ SNARK mode
int32_t otti_snark_gens_new(uint64_t nc, uint64_t nv, uint64_t ni, uint64_t nnz, otti_snark_gens **out) {
    return guarded([&] {
        if (!out) throw Error(OTTI_ERR_BAD_ARG, "null out pointer");
        auto h = std::make_unique<otti_snark_gens>(); h->g = snark_gens_new(nc, nv, ni, nnz); *out = h.release(); return OTTI_OK;
    });
}
void otti_snark_gens_free(otti_snark_gens *p) { delete p; }
int32_t otti_snark_encode(otti_instance *inst, otti_snark_gens *gens, otti_comp_comm **out) {
    return guarded([&] {
        if (!inst || !gens || !out) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        auto h = std::make_unique<otti_comp_comm>(); h->c = snark_encode_gpu(*inst->I, *gens->g); *out = h.release(); return OTTI_OK;
    });
}
int32_t otti_comp_comm_bytes(const otti_comp_comm *comm, uint8_t **out, size_t *len) {
    return guarded([&] { if (!comm || !out || !len) throw Error(OTTI_ERR_BAD_ARG, "null argument"); *out = to_malloc(comm->c->serialize(), len); return OTTI_OK; });
}
int32_t otti_comp_comm_from_bytes(const uint8_t *buf, size_t len, otti_comp_comm **out) {
    return guarded([&] {
        if (!buf || !out) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        auto h = std::make_unique<otti_comp_comm>(); h->c = CompComm::parse(buf, len); *out = h.release(); return OTTI_OK;
    });
}
void otti_comp_comm_free(otti_comp_comm *p) { delete p; }
int32_t otti_snark_prove(otti_instance *inst, otti_comp_comm *comm, const uint8_t *vars32, size_t nvars, const uint8_t *inputs32, size_t ninputs, otti_snark_gens *gens,
                         const uint8_t *tlabel, size_t tlabel_len, const uint8_t *seed32, uint32_t flags, uint8_t **proof, size_t *proof_len, double *stage_ms) {
    return guarded([&] {
        if (!inst || !comm || !gens || !proof || !proof_len) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        if (!(flags & OTTI_FLAG_GPU)) throw Error(OTTI_ERR_BAD_ARG, "OTTI_FLAG_GPU is the only proving backend; there is no CPU path");
        if (ninputs != inst->I->num_inputs) throw Error(OTTI_ERR_INVALID_NUM_INPUTS, "wrong number of inputs");
        std::vector<Fr> inputs = scalars_from_bytes(inputs32, ninputs);
        SnarkTimings tm{};
        std::vector<uint8_t> pf = snark_prove_gpu(*inst->I, *comm->c, vars32, nvars, inputs, *gens->g, tlabel, tlabel_len, seed32, &tm);
        if (stage_ms) memcpy(stage_ms, tm.ms, sizeof tm.ms);
        *proof = to_malloc(pf, proof_len); return OTTI_OK;
    });
}
int32_t otti_snark_prove_resident(otti_instance *inst, otti_comp_comm *comm, otti_witness *wit, otti_snark_gens *gens, const uint8_t *tlabel, size_t tlabel_len,
                                  const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms) {
    return guarded([&] {
        if (!inst || !comm || !wit || !gens || !proof || !proof_len) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        SnarkTimings tm{};
        std::vector<uint8_t> pf = snark_prove_resident(*inst->I, *comm->c, *wit->w, *gens->g, tlabel, tlabel_len, seed32, &tm);
        if (stage_ms) memcpy(stage_ms, tm.ms, sizeof tm.ms);
        *proof = to_malloc(pf, proof_len); return OTTI_OK;
    });
}
int32_t otti_snark_prove_sharded(otti_instance *inst, otti_comp_comm *comm, otti_witness *wit, otti_snark_gens *gens, const uint8_t *tlabel, size_t tlabel_len,
                                 const uint8_t *seed32, uint8_t **proof, size_t *proof_len, double *stage_ms) {
    return guarded([&] {
        if (!inst || !comm || !wit || !gens || !proof || !proof_len) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        if (!shard_comm()) throw Error(OTTI_ERR_BAD_ARG, "otti_shard_init has not been called");
        if (!seed32) throw Error(OTTI_ERR_BAD_ARG, "a sharded proof needs an explicit random-tape seed (the same on every rank)");
        SnarkTimings tm{};
        std::vector<uint8_t> pf = snark_prove_resident(*inst->I, *comm->c, *wit->w, *gens->g, tlabel, tlabel_len, seed32, &tm, shard_comm());
        if (stage_ms) memcpy(stage_ms, tm.ms, sizeof tm.ms);
        *proof = to_malloc(pf, proof_len); return OTTI_OK;
    });
}
int32_t otti_snark_verify(const otti_comp_comm *comm, const uint8_t *inputs32, size_t ninputs, const otti_snark_gens *gens, const uint8_t *tlabel, size_t tlabel_len,
                          const uint8_t *proof, size_t proof_len) {
    return guarded([&] {
        if (!comm || !gens || !proof) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        std::vector<Fr> inputs = scalars_from_bytes(inputs32, ninputs);
        int rc = snark_verify(*comm->c, inputs, *gens->g, tlabel, tlabel_len, proof, proof_len);
        if (rc) g_last_error = "proof rejected";
        return rc;
    });
}

// ------------------------------------------------------------------------------------------------ zkInterface / synthetic
int32_t otti_zkif_load(const char *c, const char *i, const char *w, otti_r1cs **out) {
    return guarded([&] { if (!c || !out) throw Error(OTTI_ERR_BAD_ARG, "null argument"); *out = zkif_load_impl(c, i, w); return OTTI_OK; });
}
int32_t otti_zkif_write(const otti_r1cs *r, const char *c, const char *i, const char *w) {
    return guarded([&] { if (!r || !c || !i || !w) throw Error(OTTI_ERR_BAD_ARG, "null argument"); zkif_write_impl(r, c, i, w); return OTTI_OK; });
}
void otti_r1cs_free(otti_r1cs *r) { if (!r) return; free(r->A); free(r->B); free(r->C); free(r->vars32); free(r->inputs32); free(r); }
int32_t otti_synth_r1cs(uint64_t n, uint64_t ni, uint64_t seed, otti_r1cs **out) {
    return guarded([&] {
        if (!out || n == 0) throw Error(OTTI_ERR_BAD_ARG, "bad argument");
        std::vector<otti_entry> A, B, C; std::vector<uint8_t> vars, inputs;
        synth_r1cs(n, ni, seed, A, B, C, vars, inputs);
        *out = otti_r1cs_from(n, n, ni, A, B, C, vars, inputs); return OTTI_OK;
    });
}

int32_t otti_synth_r1cs_compiler_like(uint64_t n, uint64_t ni, uint64_t seed, otti_r1cs **out) {
    return guarded([&] {
        if (!out || n < 8) throw Error(OTTI_ERR_BAD_ARG, "bad argument");
        std::vector<otti_entry> A, B, C; std::vector<uint8_t> vars, inputs;
        synth_r1cs_compiler_like(n, ni, seed, A, B, C, vars, inputs);
        *out = otti_r1cs_from(n, n, ni, A, B, C, vars, inputs); return OTTI_OK;
    });
}

// ------------------------------------------------------------------------------------------------ kernel-level entry points
namespace {
struct Staged {                      // host Montgomery bytes -> device buffer
    DevBuf<Fr> d;
    Staged(DevCtx &c, const uint8_t *h, size_t n) : d(std::max<size_t>(1, n)) { if (n) OTTI_HIP(hipMemcpyAsync(d.p, h, n * sizeof(Fr), hipMemcpyHostToDevice, c.stream)); }
};
void download(DevCtx &c, uint8_t *h, const Fr *d, size_t n) { if (n) OTTI_HIP(hipMemcpyAsync(h, d, n * sizeof(Fr), hipMemcpyDeviceToHost, c.stream)); }
struct KTimer {
    DevCtx &c; float *out;
    KTimer(DevCtx &c_, float *o) : c(c_), out(o) { if (out) OTTI_HIP(hipEventRecord(c.ev0, c.stream)); }
    void stop() { if (out) { OTTI_HIP(hipEventRecord(c.ev1, c.stream)); OTTI_HIP(hipEventSynchronize(c.ev1)); OTTI_HIP(hipEventElapsedTime(out, c.ev0, c.ev1)); } }
};
Fr fr_load(const uint8_t *p) { Fr x; memcpy(x.v, p, 32); return x; }
}  // namespace

int32_t otti_k_fr_op(int32_t op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Staged A(c, a, n), B(c, b, n); DevBuf<Fr> O(std::max<size_t>(1, n));
        KTimer t(c, ms); dev_fr_op(c, op, A.d.p, B.d.p, O.p, n); t.stop();
        download(c, out, O.p, n); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_fr_from_canonical(const uint8_t *in, uint8_t *out, size_t n) {
    return guarded([&] { DevCtx &c = DevCtx::get(); Staged A(c, in, n); dev_from_canonical(c, A.d.p, A.d.p, n); download(c, out, A.d.p, n); c.sync(); return OTTI_OK; });
}
int32_t otti_k_fr_to_canonical(const uint8_t *in, uint8_t *out, size_t n) {
    return guarded([&] { DevCtx &c = DevCtx::get(); Staged A(c, in, n); dev_to_canonical(c, A.d.p, A.d.p, n); download(c, out, A.d.p, n); c.sync(); return OTTI_OK; });
}
int32_t otti_k_multiply_vec(otti_instance *inst, const uint8_t *z, uint8_t *Az, uint8_t *Bz, uint8_t *Cz, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Instance &I = *inst->I; ensure_instance_device(I);
        Staged Z(c, z, 2 * I.num_vars); DevBuf<Fr> a(I.num_cons), b(I.num_cons), d(I.num_cons);
        KTimer t(c, ms); dev_spmv3(c, I.dev->by_row, Z.d.p, a.p, b.p, d.p, false, nullptr); t.stop();
        download(c, Az, a.p, I.num_cons); download(c, Bz, b.p, I.num_cons); download(c, Cz, d.p, I.num_cons); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_eval_table_sparse(otti_instance *inst, const uint8_t *eq_rx, const uint8_t *rABC, uint8_t *out, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Instance &I = *inst->I; ensure_instance_device(I);
        Staged E(c, eq_rx, I.num_cons); DevBuf<Fr> o(2 * I.num_vars);
        Fr coef[3] = {fr_load(rABC), fr_load(rABC + 32), fr_load(rABC + 64)};
        KTimer t(c, ms); dev_spmv3(c, I.dev->by_col, E.d.p, o.p, nullptr, nullptr, true, coef); t.stop();
        download(c, out, o.p, 2 * I.num_vars); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_eq_evals(const uint8_t *r, size_t ell, uint8_t *out, float *ms) {
    return guarded([&] {
        if (ell > 25) throw Error(OTTI_ERR_BAD_ARG, "ell > 25");
        DevCtx &c = DevCtx::get(); std::vector<Fr> rr(ell + 1); for (size_t i = 0; i < ell; i++) rr[i] = fr_load(r + 32 * i);
        size_t n = (size_t)1 << ell; DevBuf<Fr> o(n), s(5 * 4096);
        KTimer t(c, ms); dev_eq_evals(c, rr.data(), ell, o.p, s.p); t.stop();
        download(c, out, o.p, n); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_fold_top(const uint8_t *Z, size_t len, const uint8_t *r, uint8_t *out, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Staged z(c, Z, len);
        KTimer t(c, ms); dev_fold_top(c, z.d.p, len, fr_load(r)); t.stop();
        download(c, out, z.d.p, len / 2); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_fold_bot(const uint8_t *Z, size_t len, const uint8_t *r, uint8_t *out, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Staged z(c, Z, len); DevBuf<Fr> o(std::max<size_t>(1, len / 2));
        KTimer t(c, ms); dev_fold_bot(c, z.d.p, o.p, len, fr_load(r)); t.stop();
        download(c, out, o.p, len / 2); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_sc_cubic_round(const uint8_t *A, const uint8_t *B, const uint8_t *C, const uint8_t *D, size_t len, uint8_t *e3, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Staged a(c, A, len), b(c, B, len), cc(c, C, len), d(c, D, len);
        KTimer t(c, ms); auto tk = dev_sc_cubic_eval(c, a.d.p, b.d.p, cc.d.p, d.d.p, len, 0); t.stop();
        c.wait_ticket(tk); memcpy(e3, c.h_results, 96); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_sc_cubic_fold_round(const uint8_t *A, const uint8_t *B, const uint8_t *C, const uint8_t *D, size_t len, const uint8_t *r,
                                   uint8_t *out4, uint8_t *e3, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Staged a(c, A, len), b(c, B, len), cc(c, C, len), d(c, D, len);
        KTimer t(c, ms); auto tk = dev_sc_cubic_fold_eval(c, a.d.p, b.d.p, cc.d.p, d.d.p, len, fr_load(r), 0); t.stop();
        size_t h = len / 2;
        download(c, out4, a.d.p, h); download(c, out4 + 32 * h, b.d.p, h); download(c, out4 + 64 * h, cc.d.p, h); download(c, out4 + 96 * h, d.d.p, h);
        c.sync(); c.wait_ticket(tk); memcpy(e3, c.h_results, 96); return OTTI_OK;
    });
}
int32_t otti_k_sc_quad_round(const uint8_t *A, const uint8_t *B, size_t len, uint8_t *e2, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Staged a(c, A, len), b(c, B, len);
        KTimer t(c, ms); auto tk = dev_sc_quad_eval(c, a.d.p, b.d.p, len, 0); t.stop();
        c.wait_ticket(tk); memcpy(e2, c.h_results, 64); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_sc_quad_fold_round(const uint8_t *A, const uint8_t *B, size_t len, const uint8_t *r, uint8_t *out2, uint8_t *e2, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Staged a(c, A, len), b(c, B, len);
        KTimer t(c, ms); auto tk = dev_sc_quad_fold_eval(c, a.d.p, b.d.p, len, fr_load(r), 0); t.stop();
        size_t h = len / 2; download(c, out2, a.d.p, h); download(c, out2 + 32 * h, b.d.p, h);
        c.sync(); c.wait_ticket(tk); memcpy(e2, c.h_results, 64); return OTTI_OK;
    });
}
// Armed launches (device.h): the same fold + sums round three ways on the caller's tables — plain; armed and released by go() after
// `hold_us` microseconds of the kernel waiting; armed and ABORTED (the tables must come back untouched and the stream must drain),
// followed by another armed round that must still work.  out2/e2: the plain launch's folded tables and sums, for the caller's oracle.
int32_t otti_k_armed_selftest(const uint8_t *A, const uint8_t *B, size_t len, const uint8_t *r, uint32_t hold_us, uint8_t *out2, uint8_t *e2) {
    return guarded([&] {
        if (len < 8 || (len & (len - 1))) throw Error(OTTI_ERR_BAD_ARG, "table length must be a power of two >= 8");
        DevCtx &c = DevCtx::get();
        const Fr rr = fr_load(r); const size_t h = len / 2;
        Staged a0(c, A, len), b0(c, B, len), a1(c, A, len), b1(c, B, len);
        auto tk = dev_sc_quad_fold_eval(c, a0.d.p, b0.d.p, len, rr, 0);
        c.wait_ticket(tk); Fr e_plain[2] = {c.h_results[0], c.h_results[1]};
        download(c, out2, a0.d.p, h); download(c, out2 + 32 * h, b0.d.p, h); c.sync(); memcpy(e2, e_plain, 64);
        // armed, released late
        tk = dev_sc_quad_fold_eval_armed(c, a1.d.p, b1.d.p, len, 0);
        { const auto t0 = std::chrono::steady_clock::now(); while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(hold_us)) {} }
        c.go(&rr, 1);
        c.wait_ticket(tk);
        if (memcmp(e_plain, c.h_results, 64)) throw Error(OTTI_ERR_INTERNAL, "armed round: sums differ from the plain launch");
        std::vector<uint8_t> got(64 * h);
        download(c, got.data(), a1.d.p, h); download(c, got.data() + 32 * h, b1.d.p, h); c.sync();
        if (memcmp(got.data(), out2, 64 * h)) throw Error(OTTI_ERR_INTERNAL, "armed round: folded tables differ from the plain launch");
        // armed, aborted: nothing may be written, the stream must drain, the next armed launch must work
        std::vector<uint8_t> before(32 * h), after(32 * h);
        download(c, before.data(), a1.d.p, h); c.sync();
        tk = dev_sc_quad_fold_eval_armed(c, a1.d.p, b1.d.p, h, 0);
        c.go_abort();
        download(c, after.data(), a1.d.p, h); c.sync();
        if (memcmp(before.data(), after.data(), 32 * h)) throw Error(OTTI_ERR_INTERNAL, "aborted armed round wrote to its tables");
        if (*c.h_flag >= tk) throw Error(OTTI_ERR_INTERNAL, "aborted armed round delivered a result");
        tk = dev_sc_quad_fold_eval(c, a0.d.p, b0.d.p, h, rr, 0); c.wait_ticket(tk); e_plain[0] = c.h_results[0]; e_plain[1] = c.h_results[1];
        tk = dev_sc_quad_fold_eval_armed(c, a1.d.p, b1.d.p, h, 0); c.go(&rr, 1); c.wait_ticket(tk);
        if (memcmp(e_plain, c.h_results, 64)) throw Error(OTTI_ERR_INTERNAL, "armed round after an abort: sums differ from the plain launch");
        c.sync();
        // armed, and the host stalls beyond the launch's deadline (shortened to 2 ms here): the leader gives up for the WHOLE grid (nothing
        // folded, in any workgroup), says so, the host's wait fails at once, and the context is clean for the next round
        if (len >= 16) {
            const size_t q = h / 2;                                                      // both table pairs are down to q elements by now
            download(c, before.data(), a1.d.p, q); c.sync();
            struct Restore { DevCtx &c; unsigned long long d; ~Restore() { c.arm_deadline = d; } } restore{c, c.arm_deadline};
            c.arm_deadline = 200000ull;                                                  // 2 ms of the 100 MHz clock
            tk = dev_sc_quad_fold_eval_armed(c, a1.d.p, b1.d.p, q, 0);
            { const auto t0 = std::chrono::steady_clock::now(); while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(20)) {} }
            bool failed = false;
            try { c.go(&rr, 1); c.wait_ticket(tk); } catch (const Error &) { failed = true; }
            if (!failed) throw Error(OTTI_ERR_INTERNAL, "armed round past its deadline still delivered a result");
            c.arm_deadline = restore.d;
            download(c, after.data(), a1.d.p, q); c.sync();
            if (memcmp(before.data(), after.data(), 32 * q)) throw Error(OTTI_ERR_INTERNAL, "armed round past its deadline wrote to its tables");
            unsigned cnt = 1; OTTI_HIP(hipMemcpy(&cnt, c.d_counter.p, sizeof cnt, hipMemcpyDeviceToHost));
            if (cnt) throw Error(OTTI_ERR_INTERNAL, "arrival counter left non-zero after a timed-out armed round");
            tk = dev_sc_quad_fold_eval(c, a0.d.p, b0.d.p, q, rr, 0); c.wait_ticket(tk); e_plain[0] = c.h_results[0]; e_plain[1] = c.h_results[1];
            tk = dev_sc_quad_fold_eval_armed(c, a1.d.p, b1.d.p, q, 0); c.go(&rr, 1); c.wait_ticket(tk);
            if (memcmp(e_plain, c.h_results, 64)) throw Error(OTTI_ERR_INTERNAL, "armed round after a timed-out one: sums differ from the plain launch");
            c.sync();
        }
        return OTTI_OK;
    });
}
// The verifier's variable-base sum (spartan.h RowSumBeginHook / FinishHook): sum_i s[i] * decode(C[i]) on the device — batch decompression
// (k_decode_niels), LDS-bucket Pippenger (k_msm_var), window recombination on the host.  No host fallback here: this entry exists to test the device path.
int32_t otti_k_row_sum(const uint8_t *compressed32, size_t n, const uint8_t *scalars_mont32, uint8_t *out32) {
    return guarded([&] {
        if (!compressed32 || !scalars_mont32 || !out32 || n < 256) throw Error(OTTI_ERR_BAD_ARG, "null argument or fewer than 256 points");
        if (!g_row_sum_begin_hook || !g_row_sum_finish_hook) throw Error(OTTI_ERR_NO_DEVICE, "no device path registered");
        DevCtx::get();                                             // NoDevice surfaces here rather than as a declined job
        RowSumJob *job = g_row_sum_begin_hook(reinterpret_cast<const CPoint *>(compressed32), n);
        if (!job) throw Error(OTTI_ERR_NO_DEVICE, "the device declined the row sum");
        std::vector<Fr> s(n); memcpy(s.data(), scalars_mont32, 32 * n);
        Pt sum; const int rc = g_row_sum_finish_hook(job, s.data(), sum);
        if (rc == OTTI_ERR_VERIFY_DECOMPRESS) throw Error(OTTI_ERR_VERIFY_DECOMPRESS, "a point does not decode");
        if (rc) throw Error(OTTI_ERR_INTERNAL, "device row sum failed");
        pt_encode(out32, sum); return OTTI_OK;
    });
}
int32_t otti_k_msm_rows(otti_gens *gens, const uint8_t *Z, size_t L, size_t R, const uint8_t *blinds, uint8_t *out32, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Gens &g = *gens->g;
        if (R != g.R) throw Error(OTTI_ERR_BAD_ARG, "row length differs from the generator count");
        ensure_gens_device(g);
        Staged z(c, Z, L * R), bl(c, blinds, L);
        uint32_t hb = g.pc_n.h;
        const bool sparse = dev_small_fraction(c, z.d.p, L * R) > 0.25;                  // the prover takes this from the resident witness
        KTimer t(c, ms); dev_msm_rows(c, *g.dev, z.d.p, R, R, L, bl.d.p, &hb, 1, MSM_COMPRESSED, nullptr, sparse); t.stop();
        c.sync(); memcpy(out32, c.h_points, 32 * L); return OTTI_OK;
    });
}


// ---- the prover's own kernels for phase one / evaluation proof / bullet reduction
extern "C++" {
namespace {
// the two eq pyramids over m variables exactly as nizk_prove_resident lays them out (lo: last min(m,12) variables, hi: the ones before)
struct EqPyramids {
    DevBuf<Fr> buf; size_t n_lo = 0, n_hi = 0;
    EqPyramids(DevCtx &c, const Fr *tau, size_t m) : buf(8192 + 16384) {
        n_lo = std::min<size_t>(m, 12); n_hi = m - n_lo;
        dev_eq_pyramid2(c, tau + n_hi, n_lo, buf.p, tau, n_hi, n_hi ? buf.p + 8192 : nullptr);
    }
    EqSrc top() const {                                      // E over all m variables
        EqSrc e; const size_t m = n_lo + n_hi;
        if (m <= n_lo) { e.hi = nullptr; e.lo = buf.p + (((size_t)1 << m) - 1); e.lo_bits = 0; }
        else { e.hi = buf.p + 8192 + (((size_t)1 << n_hi) - 1); e.lo = buf.p + (((size_t)1 << n_lo) - 1); e.lo_bits = (int)n_lo; }
        return e;
    }
};
std::vector<Fr> fr_load_vec(const uint8_t *p, size_t n) { std::vector<Fr> v(n + 1); for (size_t i = 0; i < n; i++) v[i] = fr_load(p + 32 * i); return v; }
// Run the library's launch functions on a caller's stream for the duration of one call.  The launches share the context's scratch
// (round partials, arrival counters, MSM partials, result slots), so work enqueued on one stream must not overlap work on another:
// entering, the caller's stream waits for everything the context's own stream has been given; leaving, the context's own stream
// waits for what was just enqueued — two calls on different caller streams are thereby ordered through the context's stream.
struct StreamScope {
    DevCtx &c; hipStream_t old;
    void order(hipStream_t after, hipStream_t before) {
        if (!c.ev_order) OTTI_HIP(hipEventCreateWithFlags(&c.ev_order, hipEventDisableTiming));
        OTTI_HIP(hipEventRecord(c.ev_order, before)); OTTI_HIP(hipStreamWaitEvent(after, c.ev_order, 0));
    }
    StreamScope(DevCtx &c_, void *s) : c(c_), old(c_.stream) { if (s && (hipStream_t)s != old) { order((hipStream_t)s, old); c.stream = (hipStream_t)s; } }
    ~StreamScope() { if (c.stream != old) { hipStream_t mine = c.stream; c.stream = old; try { order(old, mine); } catch (...) {} } }
};
}  // namespace
}  // extern "C++"

int32_t otti_k_eq_pyramid(const uint8_t *r, size_t n, uint8_t *out) {
    return guarded([&] {
        if (n > 13) throw Error(OTTI_ERR_BAD_ARG, "n > 13");
        DevCtx &c = DevCtx::get(); std::vector<Fr> rr = fr_load_vec(r, n); const size_t total = ((size_t)2 << n) - 1;
        DevBuf<Fr> o(total); dev_eq_pyramid(c, rr.data(), n, o.p); download(c, out, o.p, total); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_sc_cubic3_round(const uint8_t *B, const uint8_t *C, const uint8_t *D, size_t len, const uint8_t *tau, uint8_t *e3, float *ms) {
    return guarded([&] {
        if (len < 2 || (len & (len - 1))) throw Error(OTTI_ERR_BAD_ARG, "len must be a power of two >= 2");
        DevCtx &c = DevCtx::get(); Staged b(c, B, len), cc(c, C, len), d(c, D, len);
        const size_t m = ilog2(len) - 1; std::vector<Fr> t = fr_load_vec(tau, m); EqPyramids py(c, t.data(), m);
        KTimer tm(c, ms); auto tk = dev_sc_cubic3_eval(c, b.d.p, cc.d.p, d.d.p, len, py.top(), 0); tm.stop();
        c.wait_ticket(tk); memcpy(e3, c.h_results, 96); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_sc_cubic3_fold_round(const uint8_t *B, const uint8_t *C, const uint8_t *D, size_t len, const uint8_t *r, const uint8_t *tau,
                                    uint8_t *out3, uint8_t *e3, float *ms) {
    return guarded([&] {
        if (len < 4 || (len & (len - 1))) throw Error(OTTI_ERR_BAD_ARG, "len must be a power of two >= 4");
        DevCtx &c = DevCtx::get(); Staged b(c, B, len), cc(c, C, len), d(c, D, len);
        const size_t m = ilog2(len) - 2; std::vector<Fr> t = fr_load_vec(tau, m); EqPyramids py(c, t.data(), m);
        KTimer tm(c, ms); auto tk = dev_sc_cubic3_fold_eval(c, b.d.p, cc.d.p, d.d.p, len, fr_load(r), py.top(), 0); tm.stop();
        const size_t h = len / 2;
        download(c, out3, b.d.p, h); download(c, out3 + 32 * h, cc.d.p, h); download(c, out3 + 64 * h, d.d.p, h);
        c.sync(); c.wait_ticket(tk); memcpy(e3, c.h_results, 96); return OTTI_OK;
    });
}
int32_t otti_k_poly_bound(const uint8_t *Z, size_t L, size_t R, const uint8_t *Lv, uint8_t *out, float *ms) {
    return guarded([&] {
        if (!L || !R) throw Error(OTTI_ERR_BAD_ARG, "empty matrix");
        DevCtx &c = DevCtx::get(); Staged z(c, Z, L * R), lv(c, Lv, L); DevBuf<Fr> o(R), scratch(64 * R);
        KTimer tm(c, ms); dev_poly_bound(c, z.d.p, L, R, lv.d.p, o.p, scratch.p); tm.stop();
        download(c, out, o.p, R); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_bullet_round(otti_gens *gens, size_t n_cur, int32_t fold, const uint8_t *u, const uint8_t *uinv, const uint8_t *a, const uint8_t *b,
                            const uint8_t *s, const uint8_t *blinds2, uint8_t *a_out, uint8_t *b_out, uint8_t *s_out, uint8_t *LR64, float *ms) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); Gens &g = *gens->g; ensure_gens_device(g);
        const size_t R = g.R;
        if (n_cur < 2 || n_cur > R || (n_cur & (n_cur - 1))) throw Error(OTTI_ERR_BAD_ARG, "n_cur must be a power of two in [2, R]");
        const size_t n_in = fold ? 2 * n_cur : n_cur;
        if (n_in > R) throw Error(OTTI_ERR_BAD_ARG, "folding needs 2 * n_cur <= R");
        Staged A(c, a, n_in), B(c, b, n_in), S(c, s, R); DevBuf<Fr> Ao(R), Bo(R), So(R), ex(4);
        Fr exh[4] = {fr_zero(), fr_load(blinds2), fr_zero(), fr_load(blinds2 + 32)};
        OTTI_HIP(hipMemcpyAsync(ex.p, exh, sizeof exh, hipMemcpyHostToDevice, c.stream));
        OTTI_HIP(hipMemcpyAsync(So.p, S.d.p, R * sizeof(Fr), hipMemcpyDeviceToDevice, c.stream));   // slots this round does not walk keep their value
        c.ensure_points(2, 128);
        const uint32_t qh[2] = {g.pc_1.G[0], g.pc_n.h};
        const Fr uu = fold ? fr_load(u) : fr_zero(), ui = fold ? fr_load(uinv) : fr_zero();
        KTimer tm(c, ms);
        auto tk = dev_bullet_round(c, *g.dev, R, n_cur, fold != 0, uu, ui, A.d.p, B.d.p, S.d.p, Ao.p, Bo.p, So.p, ex.p, qh);
        tm.stop();
        c.wait_points(tk); memcpy(LR64, c.h_points, 64);
        download(c, a_out, Ao.p, n_cur); download(c, b_out, Bo.p, n_cur); download(c, s_out, So.p, R); c.sync(); return OTTI_OK;
    });
}
int32_t otti_k_bullet_last_fold(size_t R, const uint8_t *u, const uint8_t *uinv, uint8_t *a2, uint8_t *b2, uint8_t *s) {
    return guarded([&] {
        if (R < 2) throw Error(OTTI_ERR_BAD_ARG, "R < 2");
        DevCtx &c = DevCtx::get(); Staged A(c, a2, 2), B(c, b2, 2), S(c, s, R); DevBuf<Fr> rows(2 * R), ex(4);
        dev_bullet_step(c, A.d.p, B.d.p, S.d.p, R, 1, true, fr_load(u), fr_load(uinv), rows.p, ex.p);
        download(c, a2, A.d.p, 1); download(c, b2, B.d.p, 1); download(c, s, S.d.p, R); c.sync(); return OTTI_OK;
    });
}

// ---- device pointers + caller's stream
static const Fr *dfr(const void *p) { return reinterpret_cast<const Fr *>(p); }
static Fr *dfr(void *p) { return reinterpret_cast<Fr *>(p); }
int32_t otti_dev_alloc(size_t nbytes, void **out) { return guarded([&] { if (!out) throw Error(OTTI_ERR_BAD_ARG, "null argument"); DevCtx::get(); OTTI_HIP(hipMalloc(out, std::max<size_t>(nbytes, 1))); return OTTI_OK; }); }
int32_t otti_dev_free(void *d) { return guarded([&] { if (d) OTTI_HIP(hipFree(d)); return OTTI_OK; }); }
int32_t otti_dev_upload(void *d, const void *h, size_t n) { return guarded([&] { DevCtx::get(); if (n) OTTI_HIP(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return OTTI_OK; }); }
int32_t otti_dev_download(void *h, const void *d, size_t n) { return guarded([&] { DevCtx::get(); if (n) OTTI_HIP(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); return OTTI_OK; }); }
int32_t otti_dev_stream_create(void **out) { return guarded([&] { if (!out) throw Error(OTTI_ERR_BAD_ARG, "null argument"); DevCtx::get(); hipStream_t s; OTTI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *out = (void *)s; return OTTI_OK; }); }
int32_t otti_dev_stream_sync(void *stream) { return guarded([&] { OTTI_HIP(hipStreamSynchronize((hipStream_t)stream)); return OTTI_OK; }); }
int32_t otti_dev_stream_destroy(void *stream) { return guarded([&] { if (stream) OTTI_HIP(hipStreamDestroy((hipStream_t)stream)); return OTTI_OK; }); }
int32_t otti_kd_multiply_vec(otti_instance *inst, const void *z, void *Az, void *Bz, void *Cz, void *stream) {
    return guarded([&] {
        if (!inst || !z || !Az || !Bz || !Cz) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        DevCtx &c = DevCtx::get(); Instance &I = *inst->I; ensure_instance_device(I); StreamScope ss(c, stream);
        dev_spmv3(c, I.dev->by_row, dfr(z), dfr(Az), dfr(Bz), dfr(Cz), false, nullptr); return OTTI_OK;
    });
}
int32_t otti_kd_eval_table_sparse(otti_instance *inst, const void *eq_rx, const uint8_t *rABC, void *out, void *stream) {
    return guarded([&] {
        if (!inst || !eq_rx || !rABC || !out) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        DevCtx &c = DevCtx::get(); Instance &I = *inst->I; ensure_instance_device(I); StreamScope ss(c, stream);
        Fr coef[3] = {fr_load(rABC), fr_load(rABC + 32), fr_load(rABC + 64)};
        dev_spmv3(c, I.dev->by_col, dfr(eq_rx), dfr(out), nullptr, nullptr, true, coef); return OTTI_OK;
    });
}
int32_t otti_kd_eq_evals(const uint8_t *r, size_t ell, void *out, void *stream) {
    return guarded([&] {
        if (ell > 25 || !out) throw Error(OTTI_ERR_BAD_ARG, "ell > 25 or null output");
        DevCtx &c = DevCtx::get(); StreamScope ss(c, stream); std::vector<Fr> rr = fr_load_vec(r, ell);
        DevBuf<Fr> s(5 * 4096);
        dev_eq_evals(c, rr.data(), ell, dfr(out), s.p);
        OTTI_HIP(hipStreamSynchronize(c.stream));                // the scratch tables are freed on return
        return OTTI_OK;
    });
}
int32_t otti_kd_fold_top(void *Z, size_t len, const uint8_t *r, void *stream) {
    return guarded([&] { DevCtx &c = DevCtx::get(); StreamScope ss(c, stream); dev_fold_top(c, dfr(Z), len, fr_load(r)); return OTTI_OK; });
}
int32_t otti_kd_fold_bot(const void *Z, void *out, size_t len, const uint8_t *r, void *stream) {
    return guarded([&] { DevCtx &c = DevCtx::get(); StreamScope ss(c, stream); dev_fold_bot(c, dfr(Z), dfr(out), len, fr_load(r)); return OTTI_OK; });
}
int32_t otti_kd_sc_cubic_round(const void *A, const void *B, const void *C, const void *D, size_t len, uint8_t *e3, void *stream) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); StreamScope ss(c, stream);
        auto tk = dev_sc_cubic_eval(c, dfr(A), dfr(B), dfr(C), dfr(D), len, 0); c.wait_ticket(tk); memcpy(e3, c.h_results, 96); return OTTI_OK;
    });
}
int32_t otti_kd_sc_cubic_fold_round(void *A, void *B, void *C, void *D, size_t len, const uint8_t *r, uint8_t *e3, void *stream) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); StreamScope ss(c, stream);
        auto tk = dev_sc_cubic_fold_eval(c, dfr(A), dfr(B), dfr(C), dfr(D), len, fr_load(r), 0); c.wait_ticket(tk); memcpy(e3, c.h_results, 96); return OTTI_OK;
    });
}
int32_t otti_kd_sc_quad_round(const void *A, const void *B, size_t len, uint8_t *e2, void *stream) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); StreamScope ss(c, stream);
        auto tk = dev_sc_quad_eval(c, dfr(A), dfr(B), len, 0); c.wait_ticket(tk); memcpy(e2, c.h_results, 64); return OTTI_OK;
    });
}
int32_t otti_kd_sc_quad_fold_round(void *A, void *B, size_t len, const uint8_t *r, uint8_t *e2, void *stream) {
    return guarded([&] {
        DevCtx &c = DevCtx::get(); StreamScope ss(c, stream);
        auto tk = dev_sc_quad_fold_eval(c, dfr(A), dfr(B), len, fr_load(r), 0); c.wait_ticket(tk); memcpy(e2, c.h_results, 64); return OTTI_OK;
    });
}
int32_t otti_kd_msm_rows(otti_gens *gens, const void *Z, size_t L, size_t R, const void *blinds, void *out32, void *stream) {
    return guarded([&] {
        if (!gens || !Z || !blinds || !out32 || !L) throw Error(OTTI_ERR_BAD_ARG, "null argument");
        DevCtx &c = DevCtx::get(); Gens &g = *gens->g;
        if (R != g.R) throw Error(OTTI_ERR_BAD_ARG, "row length differs from the generator count");
        ensure_gens_device(g); StreamScope ss(c, stream);
        uint32_t hb = g.pc_n.h;
        dev_msm_rows(c, *g.dev, dfr(Z), R, R, L, dfr(blinds), &hb, 1, MSM_COMPRESSED, nullptr, false);
        if (L > kHostEncodeRows) OTTI_HIP(hipMemcpyAsync(out32, c.d_points.p, 32 * L, hipMemcpyDeviceToDevice, c.stream));
        else { c.sync(); OTTI_HIP(hipMemcpyAsync(out32, c.h_points, 32 * L, hipMemcpyHostToDevice, c.stream)); OTTI_HIP(hipStreamSynchronize(c.stream)); }
        return OTTI_OK;
    });
}

int32_t otti_bench_madd_peak(double *madds_per_second) {
    return guarded([&] { if (!madds_per_second) throw Error(OTTI_ERR_BAD_ARG, "null argument"); *madds_per_second = dev_madd_peak(DevCtx::get()); return OTTI_OK; });
}

int32_t otti_bench_fr_mul_peak(double *products_per_second) {
    return guarded([&] { if (!products_per_second) throw Error(OTTI_ERR_BAD_ARG, "null argument"); *products_per_second = dev_fr_mul_peak(DevCtx::get()); return OTTI_OK; });
}

// ------------------------------------------------------------------------------------------------ kernel timing (HIP events on the library stream)
static const char *kClassNames[KC_COUNT] = {"msm_rows", "msm_small", "msm_finish", "sc_cubic", "sc_quad", "spmv", "eq", "reduce", "poly_bound", "bullet", "other",
                                               "pc_round", "prod_layer", "hash_layer", "gather", "dot_many", "decode", "msm_var"};
int32_t otti_stats_enable(int32_t on) { KStats::get().on = on != 0; KStats::get().mask = 0xffffffffu; KStats::get().reset(); return OTTI_OK; }
int32_t otti_stats_select(const char *kernel_class) {
    for (int k = 0; k < KC_COUNT; k++) if (!strcmp(kernel_class, kClassNames[k])) { KStats::get().mask = 1u << k; return OTTI_OK; }
    return OTTI_ERR_BAD_ARG;
}
int32_t otti_armed_launches_on(int32_t *on) { return guarded([&] { if (!on) throw Error(OTTI_ERR_BAD_ARG, "null argument"); *on = DevCtx::get().armed_ok() ? 1 : 0; return OTTI_OK; }); }
int32_t otti_stats_read(const char *kernel_class, uint64_t *count, double *total_ms) {
    return guarded([&] {
        KStats &s = KStats::get();
        if (s.used) { DevCtx::get().sync(); s.flush(); }
        for (int k = 0; k < KC_COUNT; k++) if (!strcmp(kernel_class, kClassNames[k])) { if (count) *count = s.count[k]; if (total_ms) *total_ms = s.total_ms[k]; return OTTI_OK; }
        throw Error(OTTI_ERR_BAD_ARG, "unknown kernel class");
    });
}

// ------------------------------------------------------------------------------------------------ u64-lane transport of Fr sums
void otti_lanes_pack(const uint8_t *fr, size_t n, uint64_t *lanes) {
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 8; k++) { uint32_t w; memcpy(&w, fr + 32 * i + 4 * k, 4); lanes[8 * i + k] = w; }
}
void otti_lanes_unpack(const uint64_t *lanes, size_t n, uint8_t *fr) {
    std::vector<Fr> out(n);
    lanes_to_fr(lanes, n, out.data());                                  // shard.cpp: carries, then reduction mod l (Montgomery form kept)
    for (size_t i = 0; i < n; i++) memcpy(fr + 32 * i, out[i].v, 32);
}

}  // extern "C"
