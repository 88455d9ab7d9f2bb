// spzk — drop-in for the reference's spartan-zkinterface binary.
//   spzk verify --nizk <X.zkif> <X.inp.zkif> <X.wit.zkif>      [REF /root/reference/run.py:58 (via cargo run), run.py:100 (binary)]
// Reads the three zkInterface files, builds the R1CS instance, proves it on the MI355X, verifies the proof, prints
// "Verification successful" plus stage runtimes [REF /root/reference/README.md:46-48], exit status 0 on success.
// Additive options: --seed <hex32>, --proof-out <file>, --label <transcript label>, `spzk synth <n> <prefix>` to emit a
// synthetic zkif triple.
#include <stdio.h>
#include <thread>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include "../../include/otti_spartan.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int fail(const char *what, int rc) {
    char msg[512]; otti_last_error(msg, sizeof msg);
    fprintf(stderr, "spzk: %s failed (%d): %s\n", what, rc, msg);
    return 1;
}
static int usage() {
    fprintf(stderr, "usage: spzk verify --nizk <circuit.zkif> <inputs.inp.zkif> <witness.wit.zkif> [--seed HEX64] [--proof-out FILE] [--label STR]\n"
                    "            (the reference's invocation: prove, then verify, in one process)\n"
                    "       spzk prove  --nizk <circuit.zkif> <inputs.inp.zkif> <witness.wit.zkif> --proof-out FILE [--seed HEX64] [--label STR]\n"
                    "       spzk verify --nizk <circuit.zkif> <inputs.inp.zkif> --proof-in FILE [--label STR]\n"
                    "       spzk synth <num_constraints> <out_prefix> [num_inputs] [seed]\n");
    return 2;
}

int main(int argc, char **argv) {
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
    bool nizk = false; std::vector<const char *> files; const char *seed_hex = nullptr, *proof_out = nullptr, *proof_in = nullptr, *label = "nizk_example";
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--nizk")) nizk = true;
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed_hex = argv[++i];
        else if (!strcmp(argv[i], "--proof-out") && i + 1 < argc) proof_out = argv[++i];
        else if (!strcmp(argv[i], "--proof-in") && i + 1 < argc) proof_in = argv[++i];
        else if (!strcmp(argv[i], "--label") && i + 1 < argc) label = argv[++i];
        else files.push_back(argv[i]);
    }
    if (!nizk) { fprintf(stderr, "spzk: only --nizk mode is implemented (SNARK mode is out of this path's scope)\n"); return 2; }
    const bool verify_only = proof_in != nullptr;
    if (prove_only && (verify_only || !proof_out)) return usage();
    if (verify_only ? (files.size() != 2 && files.size() != 3) : files.size() != 3) return usage();
    uint8_t seed[32]; const uint8_t *seedp = nullptr;
    if (seed_hex) {
        if (strlen(seed_hex) != 64) { fprintf(stderr, "spzk: --seed wants 64 hex digits\n"); return 2; }
        for (int i = 0; i < 32; i++) { unsigned v; if (sscanf(seed_hex + 2 * i, "%2x", &v) != 1) return usage(); seed[i] = (uint8_t)v; }
        seedp = seed;
    }
    // the HIP runtime takes a noticeable fraction of a second to come up: let it do so while the files are being parsed
    std::thread warm([&] { if (!verify_only) (void)otti_device_count(); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm};
    double t0 = now_ms();
    otti_r1cs *r = nullptr; int rc = otti_zkif_load(files[0], files[1], verify_only ? nullptr : files[2], &r); if (rc) return fail("zkif load", rc);
    double t_load = now_ms() - t0; t0 = now_ms();
    otti_instance *inst = nullptr;
    rc = otti_instance_new(r->num_cons, r->num_vars, r->num_inputs, r->A, r->nA, r->B, r->nB, r->C, r->nC, &inst); if (rc) return fail("Instance::new", rc);
    otti_gens *gens = nullptr; rc = otti_gens_new(r->num_cons, r->num_vars, r->num_inputs, &gens); if (rc) return fail("NIZKGens::new", rc);
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
    printf("* zkif_load %.3f ms\n* setup (Instance::new, NIZKGens::new, device tables) %.3f ms\n", t_load, t_setup);
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
    if (verify_only) free(proof); else otti_buf_free(proof);
    otti_gens_free(gens); otti_instance_free(inst); otti_r1cs_free(r);
    if (rc) { printf("Verification FAILED (%d: %s)\n", rc, rc == OTTI_ERR_VERIFY_DECOMPRESS ? "DecompressionError" : rc == OTTI_ERR_VERIFY_INTERNAL ? "InternalError" : "malformed proof"); return 1; }
    if (wrc) return 1;
    if (prove_only) { printf("Proof written to %s (%zu bytes)\n", proof_out, proof_len); return 0; }
    printf("Verification successful\n");
    return 0;
}
