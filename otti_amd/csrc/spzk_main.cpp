// spzk — drop-in for the reference's spartan-zkinterface binary.
//   spzk verify --nizk <X.zkif> <X.inp.zkif> <X.wit.zkif>      [REF /root/reference/run.py:58 (via cargo run), run.py:100 (binary)]
// Reads the three zkInterface files, builds the R1CS instance, proves it on the MI355X, verifies the proof, prints
// "Verification successful" plus stage runtimes [REF /root/reference/README.md:46-48], exit status 0 on success.
// Without --nizk the same files go through SNARK mode (upstream spartan-zkinterface's default [RECALL]): SNARK::encode commits to the
// circuit, SNARK::prove adds the R1CSEvalProof, SNARK::verify checks it against the commitment alone.
// Additive options: --seed <hex32>, --proof-out <file>, --label <transcript label>, `spzk synth <n> <prefix>` to emit a
// synthetic zkif triple.
#include <stdio.h>
#include <thread>
#include <time.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <algorithm>
#include <unistd.h>
#include "../../include/otti_spartan.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int fail(const char *what, int rc) {
    char msg[512]; otti_last_error(msg, sizeof msg);
    fprintf(stderr, "spzk: %s failed (%d): %s\n", what, rc, msg);
    return 1;
}
static int usage() {
    fprintf(stderr, "usage: spzk verify [--nizk] <circuit.zkif> <inputs.inp.zkif> <witness.wit.zkif> [--seed HEX64] [--proof-out FILE] [--label STR]\n"
                    "            (the reference's invocation: prove, then verify, in one process; without --nizk: SNARK mode)\n"
                    "       spzk prove  --nizk <circuit.zkif> <inputs.inp.zkif> <witness.wit.zkif> --proof-out FILE [--seed HEX64] [--label STR]\n"
                    "       spzk verify --nizk <circuit.zkif> <inputs.inp.zkif> --proof-in FILE [--label STR]\n"
                    "       spzk synth <num_constraints> <out_prefix> [num_inputs] [seed]\n");
    return 2;
}

// milliseconds from the creation of this process (execve) to now: /proc/self/stat's start time (clock ticks since boot) against
// CLOCK_BOOTTIME — what the dynamic loader spent mapping libamdhip64 and its dependencies before main() ran (10 ms resolution)
static double ms_since_process_start() {
    FILE *f = fopen("/proc/self/stat", "r"); if (!f) return -1;
    char buf[1024]; size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[n] = 0;
    const char *p = strrchr(buf, ')'); if (!p) return -1;
    unsigned long long start = 0; int field = 2;                  // p points at the end of field 2 (comm)
    for (p++; *p && field < 22; p++) if (*p == ' ') { field++; if (field == 22) { start = strtoull(p + 1, nullptr, 10); break; } }
    if (!start) return -1;
    struct timespec ts; if (clock_gettime(CLOCK_BOOTTIME, &ts) != 0) return -1;
    return (ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6) - 1e3 * (double)start / (double)sysconf(_SC_CLK_TCK);
}

int main(int argc, char **argv) {
    const double t_before_main = ms_since_process_start(), t_main0 = now_ms();
    if (argc < 2) return usage();
    if (!strcmp(argv[1], "synth")) {
        if (argc < 4) return usage();
        uint64_t n = strtoull(argv[2], 0, 0), ni = argc > 4 ? strtoull(argv[4], 0, 0) : 10, seed = argc > 5 ? strtoull(argv[5], 0, 0) : 1;
        otti_r1cs *r = nullptr; int rc = otti_synth_r1cs(n, ni, seed, &r); if (rc) return fail("synth", rc);
        std::string p = argv[3];
        rc = otti_zkif_write(r, (p + ".zkif").c_str(), (p + ".inp.zkif").c_str(), (p + ".wit.zkif").c_str());
        otti_r1cs_free(r); if (rc) return fail("zkif write", rc);
        printf("wrote %s.zkif %s.inp.zkif %s.wit.zkif (%llu constraints)\n", p.c_str(), p.c_str(), p.c_str(), (unsigned long long)n);
        return 0;
    }
    const bool prove_only = !strcmp(argv[1], "prove");
    if (!prove_only && strcmp(argv[1], "verify")) return usage();
    // one proof per process: the generator window table is built and used once, so a narrow window (small table, ~7 ms to build for
    // R = 1024) beats the wide one a long-lived prover process amortises (see prover.cpp device_window_bits); an explicit setting wins
    setenv("OTTI_MSM_WINDOW", "10", 0);
    bool nizk = false, label_given = false; std::vector<const char *> files; const char *seed_hex = nullptr, *proof_out = nullptr, *proof_in = nullptr, *label = "nizk_example";
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--nizk")) nizk = true;
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed_hex = argv[++i];
        else if (!strcmp(argv[i], "--proof-out") && i + 1 < argc) proof_out = argv[++i];
        else if (!strcmp(argv[i], "--proof-in") && i + 1 < argc) proof_in = argv[++i];
        else if (!strcmp(argv[i], "--label") && i + 1 < argc) { label = argv[++i]; label_given = true; }
        else files.push_back(argv[i]);
    }
    const bool verify_only = proof_in != nullptr;
    if (!nizk && (prove_only || verify_only)) { fprintf(stderr, "spzk: separate prove / verify processes are offered in --nizk mode only\n"); return 2; }
    if (!nizk && !label_given) label = "snark_example";
    if (prove_only && (verify_only || !proof_out)) return usage();
    if (verify_only ? (files.size() != 2 && files.size() != 3) : files.size() != 3) return usage();
    uint8_t seed[32]; const uint8_t *seedp = nullptr;
    if (seed_hex) {
        if (strlen(seed_hex) != 64) { fprintf(stderr, "spzk: --seed wants 64 hex digits\n"); return 2; }
        for (int i = 0; i < 32; i++) { unsigned v; if (sscanf(seed_hex + 2 * i, "%2x", &v) != 1) return usage(); seed[i] = (uint8_t)v; }
        seedp = seed;
    }
    // the HIP runtime takes a noticeable fraction of a second to come up: let it do so while the files are being parsed
    std::thread warm([&] { if (!verify_only && otti_device_count() > 0) (void)otti_prepare_device(nullptr, nullptr); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm};
    double t0 = now_ms();
    otti_r1cs *r = nullptr; int rc = otti_zkif_load(files[0], files[1], verify_only ? nullptr : files[2], &r); if (rc) return fail("zkif load", rc);
    double t_load = now_ms() - t0; t0 = now_ms();
    // NIZKGens::new (a thousand hash-to-group maps and the host window tables) depends on the sizes only: it runs next to Instance::new
    otti_gens *gens_early = nullptr; int gens_rc = 0;
    std::thread gens_thread;
    if (nizk) gens_thread = std::thread([&] { gens_rc = otti_gens_new(r->num_cons, r->num_vars, r->num_inputs, &gens_early); });
    auto gens_rc_wait = [&] { if (gens_thread.joinable()) gens_thread.join(); return gens_rc; };
    struct GensJoiner { std::thread &t; ~GensJoiner() { if (t.joinable()) t.join(); } } gens_joiner{gens_thread};
    otti_instance *inst = nullptr;
    rc = otti_instance_new(r->num_cons, r->num_vars, r->num_inputs, r->A, r->nA, r->B, r->nB, r->C, r->nC, &inst); if (rc) return fail("Instance::new", rc);
    if (!nizk) {                                               // ---- SNARK mode: encode, prove, verify
        const uint64_t nnz = std::max<uint64_t>({(uint64_t)r->nA, (uint64_t)r->nB, (uint64_t)r->nC});
        otti_snark_gens *sg = nullptr; rc = otti_snark_gens_new(r->num_cons, r->num_vars, r->num_inputs, nnz, &sg); if (rc) return fail("SNARKGens::new", rc);
        double t_setup = now_ms() - t0; t0 = now_ms();
        otti_comp_comm *cc = nullptr; rc = otti_snark_encode(inst, sg, &cc); if (rc) return fail("SNARK::encode", rc);
        double t_encode = now_ms() - t0; t0 = now_ms();
        uint8_t *proof = nullptr; size_t proof_len = 0; double st[10] = {0};
        rc = otti_snark_prove(inst, cc, r->vars32, r->nvars, r->inputs32, r->ninputs, sg, (const uint8_t *)label, strlen(label), seedp, OTTI_FLAG_GPU, &proof, &proof_len, st);
        if (rc) return fail("SNARK::prove", rc);
        double t_prove = now_ms() - t0; t0 = now_ms();
        // the verifier's copy of the commitment: its bytes only
        uint8_t *cb = nullptr; size_t cb_len = 0; rc = otti_comp_comm_bytes(cc, &cb, &cb_len); if (rc) return fail("commitment bytes", rc);
        otti_comp_comm *vc = nullptr; rc = otti_comp_comm_from_bytes(cb, cb_len, &vc); if (rc) return fail("commitment parse", rc);
        rc = otti_snark_verify(vc, r->inputs32, r->ninputs, sg, (const uint8_t *)label, strlen(label), proof, proof_len);
        double t_verify = now_ms() - t0;
        uint64_t nc, nv, ni; otti_instance_dims(inst, &nc, &nv, &ni);
        printf("* instance: %llu constraints (padded %llu), %llu variables (padded %llu), %llu inputs, %llu non-zero entries in the largest matrix\n", (unsigned long long)r->num_cons,
               (unsigned long long)nc, (unsigned long long)r->num_vars, (unsigned long long)nv, (unsigned long long)ni, (unsigned long long)nnz);
        printf("* zkif_load %.3f ms\n* setup (Instance::new, SNARKGens::new) %.3f ms\n* SNARK::encode %.3f ms (computation commitment %zu bytes)\n", t_load, t_setup, t_encode, cb_len);
        printf("* SNARK::prove %.3f ms\n  * polycommit %.3f ms\n  * multiply_vec %.3f ms\n  * prove_sc_phase_one %.3f ms\n  * eval_table_sparse %.3f ms\n  * prove_sc_phase_two %.3f ms\n"
               "  * polyeval %.3f ms\n  * R1CSEvalProof: derefs commitment %.3f ms, product circuits %.3f ms, hash layer %.3f ms\n  * len_snark_proof %zu\n",
               t_prove, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], proof_len);
        printf("* SNARK::verify %.3f ms\n", t_verify);
        int wrc = 0;
        if (proof_out && !rc) { FILE *f = fopen(proof_out, "wb"); if (!f || fwrite(proof, 1, proof_len, f) != proof_len) { fprintf(stderr, "spzk: cannot write %s\n", proof_out); wrc = 1; } if (f) fclose(f); }
        otti_buf_free(proof); otti_buf_free(cb); otti_comp_comm_free(vc); otti_comp_comm_free(cc); otti_snark_gens_free(sg); otti_instance_free(inst); otti_r1cs_free(r);
        if (rc) { printf("Verification FAILED (%d)\n", rc); return 1; }
        if (wrc) return 1;
        printf("Verification successful\n");
        return 0;
    }
    otti_gens *gens = nullptr; rc = gens_rc_wait(); if (rc) return fail("NIZKGens::new", rc);
    gens = gens_early;
    double t_objs = now_ms() - t0;
    if (!verify_only) { rc = otti_prepare_device(inst, gens); if (rc) return fail("device setup", rc); }
    double t_setup = now_ms() - t0; t0 = now_ms();
    uint8_t *proof = nullptr; size_t proof_len = 0; double st[8] = {0}, t_prove = 0, t_verify = 0;
    if (verify_only) {
        FILE *f = fopen(proof_in, "rb");
        if (!f) { fprintf(stderr, "spzk: cannot read %s\n", proof_in); return 1; }
        fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
        proof = (uint8_t *)malloc(len > 0 ? (size_t)len : 1); proof_len = len > 0 ? (size_t)len : 0;
        if (fread(proof, 1, proof_len, f) != proof_len) { fclose(f); fprintf(stderr, "spzk: short read on %s\n", proof_in); return 1; }
        fclose(f);
    } else {
        rc = otti_nizk_prove(inst, r->vars32, r->nvars, r->inputs32, r->ninputs, gens, (const uint8_t *)label, strlen(label), seedp, OTTI_FLAG_GPU, &proof,
                             &proof_len, st);
        if (rc) return fail("NIZK::prove", rc);
        t_prove = now_ms() - t0; t0 = now_ms();
    }
    if (!prove_only) {
        rc = otti_nizk_verify(inst, r->inputs32, r->ninputs, gens, (const uint8_t *)label, strlen(label), proof, proof_len);
        t_verify = now_ms() - t0;
    }
    uint64_t nc, nv, ni; otti_instance_dims(inst, &nc, &nv, &ni);
    printf("* instance: %llu constraints (padded %llu), %llu variables (padded %llu), %llu inputs\n", (unsigned long long)r->num_cons, (unsigned long long)nc,
           (unsigned long long)r->num_vars, (unsigned long long)nv, (unsigned long long)ni);
    printf("* zkif_load %.3f ms\n* setup (Instance::new, NIZKGens::new, device tables) %.3f ms\n  * host objects %.3f ms\n", t_load, t_setup, t_objs);
    if (!verify_only)
        printf("* NIZK::prove %.3f ms\n  * polycommit %.3f ms\n  * multiply_vec %.3f ms\n  * prove_sc_phase_one %.3f ms\n  * eval_table_sparse %.3f ms\n"
               "  * prove_sc_phase_two %.3f ms\n  * polyeval %.3f ms\n  * len_r1cs_sat_proof %zu\n", t_prove, st[0], st[1], st[2], st[3], st[4], st[5], proof_len);
    if (!prove_only) printf("* NIZK::verify %.3f ms\n", t_verify);
    int wrc = 0;
    if (proof_out && !rc) {
        FILE *f = fopen(proof_out, "wb");
        if (!f || fwrite(proof, 1, proof_len, f) != proof_len) { fprintf(stderr, "spzk: cannot write %s\n", proof_out); wrc = 1; }
        if (f) fclose(f);
    }
    // (no frees: the process ends here and hands everything back at once — releasing the device tables one by one costs milliseconds)
    printf("* process start to main() %.0f ms (the dynamic loader mapping the HIP runtime), main() %.3f ms\n", t_before_main, now_ms() - t_main0);
    if (rc) { printf("Verification FAILED (%d: %s)\n", rc, rc == OTTI_ERR_VERIFY_DECOMPRESS ? "DecompressionError" : rc == OTTI_ERR_VERIFY_INTERNAL ? "InternalError" : "malformed proof"); return 1; }
    if (wrc) return 1;
    if (prove_only) { printf("Proof written to %s (%zu bytes)\n", proof_out, proof_len); fflush(stdout); _exit(0); }
    printf("Verification successful\n");
    fflush(stdout);
    _exit(0);                                                  // everything is printed and written: skip the HIP runtime's teardown
}
