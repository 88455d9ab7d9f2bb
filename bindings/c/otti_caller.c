/*
 * In-process caller of libottispartan through include/otti_spartan.h, in plain C: the compiled, tested stand-in for the Rust shim
 * crate of bindings/rust/otti-spartan (this image has no rustc), mirroring it call for call — Instance::new, VarsAssignment::new /
 * InputsAssignment::new (canonical check), Instance::is_sat, NIZKGens::new, NIZK::prove, NIZK::verify, error mapping — i.e. what
 * rust-circ `--action spartan` does with libspartan in-process [REF /root/reference/run.py:147, Dockerfile:35-37].
 *
 *   otti_caller host  <proof-in>            no GPU needed: builds the synthetic 2^8 instance, checks is_sat, the R1CSError mapping,
 *                                           that NIZK::prove reports "no device" cleanly when there is none, and verifies the proof
 *                                           file (made by the CPU oracle in the test) plus a tampered copy
 *   otti_caller prove <log2 n> <proof-out>  needs the MI355X: prove (seeded), verify, reject a tampered copy, write the proof
 *
 * SNARK mode — what upstream spartan-zkinterface runs WITHOUT --nizk: SNARKGens::new, SNARK::encode, SNARK::prove, SNARK::verify:
 *   otti_caller snark-host <comm-in> <proof-in>          no GPU needed: ComputationCommitment from bytes, SNARK::verify of a proof made
 *                                                        elsewhere (the CPU oracle's, in the test), a tampered copy, another label
 *   otti_caller snark <log2 n> <comm-out> <proof-out>    needs the MI355X: encode, prove (seeded), verify with the verifier's copy of the
 *                                                        commitment (bytes only), reject a tampered proof, write commitment and proof
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "otti_spartan.h"

/* ---- the shim's types */
typedef struct { otti_instance *h; } Instance;
typedef struct { otti_gens *h; } NIZKGens;
typedef struct { uint8_t (*bytes)[32]; size_t n; } Assignment;           /* VarsAssignment = InputsAssignment = Assignment */
typedef struct { uint8_t *bytes; size_t len; } NIZK;
typedef enum { R1CS_OK = 0, NonPowerOfTwoCons = -1, NonPowerOfTwoVars = -2, InvalidNumberOfInputs = -3, InvalidNumberOfVars = -4,
               InvalidScalar = -5, InvalidIndex = -6 } R1CSError;
typedef enum { VERIFY_OK = 0, InternalError = -10, DecompressionError = -11 } ProofVerifyError;

static void last_error(char *buf, size_t cap) { otti_last_error(buf, cap); }
static int canonical(const uint8_t s[32]) {
    static const uint8_t L[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
    for (int i = 31; i >= 0; i--) { if (s[i] < L[i]) return 1; if (s[i] > L[i]) return 0; }
    return 0;
}
static R1CSError Instance_new(size_t nc, size_t nv, size_t ni, const otti_entry *A, size_t nA, const otti_entry *B, size_t nB, const otti_entry *C,
                              size_t nC, Instance *out) {
    return (R1CSError)otti_instance_new(nc, nv, ni, A, nA, B, nB, C, nC, &out->h);
}
static R1CSError Assignment_new(const uint8_t (*a)[32], size_t n, Assignment *out) {
    for (size_t i = 0; i < n; i++) if (!canonical(a[i])) return InvalidScalar;
    out->bytes = malloc(n ? 32 * n : 1); memcpy(out->bytes, a, 32 * n); out->n = n; return R1CS_OK;
}
static R1CSError Instance_is_sat(const Instance *I, const Assignment *vars, const Assignment *inputs, int *sat) {
    int32_t s = 0; int32_t rc = otti_instance_is_sat(I->h, (const uint8_t *)vars->bytes, vars->n, (const uint8_t *)inputs->bytes, inputs->n, &s);
    *sat = s; return (R1CSError)rc;
}
static int NIZKGens_new(size_t nc, size_t nv, size_t ni, NIZKGens *out) { return otti_gens_new(nc, nv, ni, &out->h); }
/* NIZK::prove(&inst, vars, &inputs, &gens, &mut Transcript::new(label)); seed NULL = OS entropy like upstream's OsRng */
static int NIZK_prove(const Instance *I, const Assignment *vars, const Assignment *inputs, const NIZKGens *g, const char *label, const uint8_t *seed32, NIZK *out) {
    return otti_nizk_prove(I->h, (const uint8_t *)vars->bytes, vars->n, (const uint8_t *)inputs->bytes, inputs->n, g->h, (const uint8_t *)label, strlen(label),
                           seed32, OTTI_FLAG_GPU, &out->bytes, &out->len, NULL);
}
static ProofVerifyError NIZK_verify(const NIZK *p, const Instance *I, const Assignment *inputs, const char *label, const NIZKGens *g) {
    int32_t rc = otti_nizk_verify(I->h, (const uint8_t *)inputs->bytes, inputs->n, g->h, (const uint8_t *)label, strlen(label), p->bytes, p->len);
    return rc == 0 ? VERIFY_OK : rc == OTTI_ERR_VERIFY_DECOMPRESS ? DecompressionError : InternalError;
}

/* ---- SNARK mode: upstream lib.rs SNARKGens / ComputationCommitment (+ ComputationDecommitment) / SNARK */
typedef struct { otti_snark_gens *h; } SNARKGens;
typedef struct { otti_comp_comm *h; } ComputationCommitment;           /* made by encode: carries the decommitment SNARK::prove needs */
typedef NIZK SNARK;
static int SNARKGens_new(size_t nc, size_t nv, size_t ni, size_t num_nz_entries, SNARKGens *out) { return otti_snark_gens_new(nc, nv, ni, num_nz_entries, &out->h); }
/* let (comm, decomm) = SNARK::encode(&inst, &gens); */
static int SNARK_encode(const Instance *I, const SNARKGens *g, ComputationCommitment *out) { return otti_snark_encode(I->h, g->h, &out->h); }
static int ComputationCommitment_from_bytes(const uint8_t *b, size_t n, ComputationCommitment *out) { return otti_comp_comm_from_bytes(b, n, &out->h); }
/* SNARK::prove(&inst, &comm, &decomm, vars, &inputs, &gens, &mut Transcript::new(label)) */
static int SNARK_prove(const Instance *I, const ComputationCommitment *comm, const Assignment *vars, const Assignment *inputs, const SNARKGens *g, const char *label,
                       const uint8_t *seed32, SNARK *out) {
    return otti_snark_prove(I->h, comm->h, (const uint8_t *)vars->bytes, vars->n, (const uint8_t *)inputs->bytes, inputs->n, g->h, (const uint8_t *)label, strlen(label),
                            seed32, OTTI_FLAG_GPU, &out->bytes, &out->len, NULL);
}
/* proof.verify(&comm, &inputs, &mut Transcript::new(label), &gens): the commitment's bytes are all the verifier holds of the circuit */
static ProofVerifyError SNARK_verify(const SNARK *p, const ComputationCommitment *comm, const Assignment *inputs, const char *label, const SNARKGens *g) {
    int32_t rc = otti_snark_verify(comm->h, (const uint8_t *)inputs->bytes, inputs->n, g->h, (const uint8_t *)label, strlen(label), p->bytes, p->len);
    return rc == 0 ? VERIFY_OK : rc == OTTI_ERR_VERIFY_DECOMPRESS ? DecompressionError : InternalError;
}

#define CHECK(cond, what) do { if (!(cond)) { char m[256]; last_error(m, sizeof m); fprintf(stderr, "otti_caller: %s (%s)\n", what, m); return 1; } } while (0)

static int load_file(const char *path, NIZK *p) {
    FILE *f = fopen(path, "rb"); if (!f) return 0;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    p->bytes = malloc(n > 0 ? (size_t)n : 1); p->len = (size_t)n;
    int ok = fread(p->bytes, 1, p->len, f) == p->len; fclose(f); return ok;
}

static int write_file(const char *path, const uint8_t *b, size_t n) { FILE *f = fopen(path, "wb"); if (!f) return 0; int ok = fwrite(b, 1, n, f) == n; fclose(f); return ok; }

/* `snark-host <comm> <proof>` and `snark <log2 n> <comm-out> <proof-out>` */
static int snark_main(int argc, char **argv) {
    const int host_only = !strcmp(argv[1], "snark-host");
    if (argc < (host_only ? 4 : 5)) { fprintf(stderr, "usage: otti_caller snark-host <comm-in> <proof-in> | snark <log2 n> <comm-out> <proof-out>\n"); return 2; }
    const size_t n = host_only ? 256 : (size_t)1 << atoi(argv[2]), ni = 10;
    const char *label = "snark_example";
    uint8_t seed[32]; memset(seed, 0x2a, 32);
    otti_r1cs *r = NULL;
    CHECK(otti_synth_r1cs(n, ni, 1, &r) == 0, "synthetic instance");
    Instance inst; SNARKGens gens; Assignment vars, inputs;
    CHECK(Instance_new(r->num_cons, r->num_vars, r->num_inputs, r->A, r->nA, r->B, r->nB, r->C, r->nC, &inst) == R1CS_OK, "Instance::new");
    CHECK(Assignment_new((const uint8_t (*)[32])r->vars32, r->nvars, &vars) == R1CS_OK, "VarsAssignment::new");
    CHECK(Assignment_new((const uint8_t (*)[32])r->inputs32, r->ninputs, &inputs) == R1CS_OK, "InputsAssignment::new");
    size_t nz = r->nA > r->nB ? r->nA : r->nB; if (r->nC > nz) nz = r->nC;             /* what spartan-zkinterface passes: the largest matrix */
    CHECK(SNARKGens_new(r->num_cons, r->num_vars, r->num_inputs, nz, &gens) == 0, "SNARKGens::new");
    SNARK proof = {0, 0}; NIZK comm_bytes = {0, 0}; ComputationCommitment verifier_comm;
    if (host_only) {
        if (otti_device_count() == 0) {
            ComputationCommitment none; CHECK(SNARK_encode(&inst, &gens, &none) == OTTI_ERR_NO_DEVICE, "SNARK::encode without a device must return OTTI_ERR_NO_DEVICE (no CPU fallback)");
        }
        CHECK(load_file(argv[2], &comm_bytes) && load_file(argv[3], &proof), "commitment / proof file");
    } else {
        ComputationCommitment comm;
        CHECK(SNARK_encode(&inst, &gens, &comm) == 0, "SNARK::encode");
        CHECK(otti_comp_comm_bytes(comm.h, &comm_bytes.bytes, &comm_bytes.len) == 0, "commitment bytes");
        CHECK(SNARK_prove(&inst, &comm, &vars, &inputs, &gens, label, seed, &proof) == 0, "SNARK::prove");
        CHECK(write_file(argv[3], comm_bytes.bytes, comm_bytes.len) && write_file(argv[4], proof.bytes, proof.len), "write commitment / proof");
        otti_comp_comm_free(comm.h);
    }
    CHECK(ComputationCommitment_from_bytes(comm_bytes.bytes, comm_bytes.len, &verifier_comm) == 0, "ComputationCommitment from bytes");
    CHECK(SNARK_verify(&proof, &verifier_comm, &inputs, label, &gens) == VERIFY_OK, "SNARK::verify");
    CHECK(SNARK_verify(&proof, &verifier_comm, &inputs, "another label", &gens) != VERIFY_OK, "a different transcript label must not verify");
    {
        SNARK t = {malloc(proof.len), proof.len}; memcpy(t.bytes, proof.bytes, proof.len);
        t.bytes[(2 * proof.len) / 3] ^= 0x40;
        CHECK(SNARK_verify(&t, &verifier_comm, &inputs, label, &gens) != VERIFY_OK, "a tampered proof must not verify");
        free(t.bytes);
    }
    {
        ComputationCommitment bad; CHECK(ComputationCommitment_from_bytes(comm_bytes.bytes, comm_bytes.len / 2, &bad) != 0, "a truncated commitment must not parse");
    }
    printf("otti_caller ok: SNARK mode, %zu constraints, commitment %zu bytes, proof %zu bytes\n", n, comm_bytes.len, proof.len);
    if (host_only) { free(comm_bytes.bytes); free(proof.bytes); } else { otti_buf_free(comm_bytes.bytes); otti_buf_free(proof.bytes); }
    otti_comp_comm_free(verifier_comm.h); free(vars.bytes); free(inputs.bytes); otti_snark_gens_free(gens.h); otti_instance_free(inst.h); otti_r1cs_free(r);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: otti_caller host <proof-in> | prove <log2 n> <proof-out> | snark-host <comm-in> <proof-in> | snark <log2 n> <comm-out> <proof-out>\n"); return 2; }
    if (!strncmp(argv[1], "snark", 5)) return snark_main(argc, argv);
    const int host_only = !strcmp(argv[1], "host");
    const size_t n = host_only ? 256 : (size_t)1 << atoi(argv[2]), ni = 10;
    const char *label = "nizk_example";
    uint8_t seed[32]; memset(seed, 0x2a, 32);
    otti_r1cs *r = NULL;
    CHECK(otti_synth_r1cs(n, ni, 1, &r) == 0, "synthetic instance");
    Instance inst; NIZKGens gens; Assignment vars, inputs;
    CHECK(Instance_new(r->num_cons, r->num_vars, r->num_inputs, r->A, r->nA, r->B, r->nB, r->C, r->nC, &inst) == R1CS_OK, "Instance::new");
    CHECK(Assignment_new((const uint8_t (*)[32])r->vars32, r->nvars, &vars) == R1CS_OK, "VarsAssignment::new");
    CHECK(Assignment_new((const uint8_t (*)[32])r->inputs32, r->ninputs, &inputs) == R1CS_OK, "InputsAssignment::new");
    int sat = 0;
    CHECK(Instance_is_sat(&inst, &vars, &inputs, &sat) == R1CS_OK && sat, "Instance::is_sat");
    CHECK(NIZKGens_new(r->num_cons, r->num_vars, r->num_inputs, &gens) == 0, "NIZKGens::new");
    /* error mapping, as upstream's tests in lib.rs: bad index, bad scalar, wrong number of inputs, unsatisfying assignment */
    {
        Instance bad; otti_entry e = r->A[0]; e.col = r->num_vars + r->num_inputs + 1;
        CHECK(Instance_new(r->num_cons, r->num_vars, r->num_inputs, &e, 1, r->B, r->nB, r->C, r->nC, &bad) == InvalidIndex, "InvalidIndex expected");
        e = r->A[0]; memset(e.val, 0xff, 32);
        CHECK(Instance_new(r->num_cons, r->num_vars, r->num_inputs, &e, 1, r->B, r->nB, r->C, r->nC, &bad) == InvalidScalar, "InvalidScalar expected");
        uint8_t ff[1][32]; memset(ff, 0xff, 32); Assignment a;
        CHECK(Assignment_new((const uint8_t (*)[32])ff, 1, &a) == InvalidScalar, "InvalidScalar expected from the assignment");
        Assignment few = inputs; few.n = ni - 1; int s2 = 0;
        CHECK(Instance_is_sat(&inst, &vars, &few, &s2) == InvalidNumberOfInputs, "InvalidNumberOfInputs expected");
        Assignment wrong; Assignment_new((const uint8_t (*)[32])r->vars32, r->nvars, &wrong); wrong.bytes[3][0] ^= 1;
        CHECK(Instance_is_sat(&inst, &wrong, &inputs, &s2) == R1CS_OK && !s2, "unsatisfying assignment must be reported");
        free(wrong.bytes);
    }
    NIZK proof = {0, 0};
    if (host_only) {
        if (otti_device_count() == 0) {
            NIZK none = {0, 0}; int rc = NIZK_prove(&inst, &vars, &inputs, &gens, label, seed, &none);
            CHECK(rc == OTTI_ERR_NO_DEVICE, "NIZK::prove without a device must return OTTI_ERR_NO_DEVICE (no CPU fallback)");
        }
        CHECK(load_file(argv[2], &proof), "proof file");
    } else {
        CHECK(argc >= 4, "prove needs an output path");
        CHECK(NIZK_prove(&inst, &vars, &inputs, &gens, label, seed, &proof) == 0, "NIZK::prove");
    }
    CHECK(NIZK_verify(&proof, &inst, &inputs, label, &gens) == VERIFY_OK, "NIZK::verify");
    CHECK(NIZK_verify(&proof, &inst, &inputs, "another label", &gens) != VERIFY_OK, "a different transcript label must not verify");
    {
        NIZK t = {malloc(proof.len), proof.len}; memcpy(t.bytes, proof.bytes, proof.len);
        t.bytes[proof.len / 2] ^= 0x40;
        CHECK(NIZK_verify(&t, &inst, &inputs, label, &gens) != VERIFY_OK, "a tampered proof must not verify");
        free(t.bytes);
    }
    if (!host_only) {
        FILE *f = fopen(argv[3], "wb"); CHECK(f && fwrite(proof.bytes, 1, proof.len, f) == proof.len, "write proof"); fclose(f);
        otti_buf_free(proof.bytes);
    } else free(proof.bytes);
    printf("otti_caller ok: %zu constraints, proof %zu bytes\n", n, proof.len);
    free(vars.bytes); free(inputs.bytes); otti_gens_free(gens.h); otti_instance_free(inst.h); otti_r1cs_free(r);
    return 0;
}
