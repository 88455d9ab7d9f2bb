// AddressSanitizer / UBSan harness for the host-side parsers of untrusted input (CPU build only: GPU sanitizers are not available on
// the pool).  Compiles the product's host sources directly (no HIP, no device code) and feeds them mutated .zkif files and proofs.
// usage: san_harness <circuit.zkif> <inputs.zkif> <witness.zkif> <proof.bin> <label> <workdir> <iterations> [<snark_comm.bin> <snark_proof.bin> <nnz>]
// (with the last three: SNARK::verify of the given and of mutated proofs as well).  The same source builds with -fsanitize=thread (make tsan): the
// verifiers hand their group equations to background threads through lock-free slots (snark.h Deferred), which is what that build checks.
#include "../../otti_amd/csrc/spartan.h"
#include "../../otti_amd/csrc/snark.h"
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

using namespace otti;
otti_r1cs *zkif_load_impl(const char *circuit_path, const char *inputs_path, const char *witness_path);
extern "C" void otti_r1cs_free(otti_r1cs *r) { if (!r) return; free(r->A); free(r->B); free(r->C); free(r->vars32); free(r->inputs32); free(r); }

static uint64_t rng_state = 0x243f6a8885a308d3ULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static std::vector<uint8_t> slurp(const char *p) {
    FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    std::vector<uint8_t> d; uint8_t b[4096]; size_t k; while ((k = fread(b, 1, sizeof b, f)) > 0) d.insert(d.end(), b, b + k); fclose(f); return d;
}
static void spit(const std::string &p, const std::vector<uint8_t> &d) { FILE *f = fopen(p.c_str(), "wb"); if (!f) { perror(p.c_str()); exit(2); } fwrite(d.data(), 1, d.size(), f); fclose(f); }
static std::vector<uint8_t> mutate(std::vector<uint8_t> b) {
    switch (rnd() % 5) {
    case 0: if (b.size() > 1) b.resize(rnd() % b.size()); break;
    case 1: for (int k = 0, n = 1 + rnd() % 5; k < n && !b.empty(); k++) b[rnd() % b.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
    case 2: if (b.size() > 4) { static const uint32_t ext[6] = {0, 1, 0x7fffffffu, 0xffffffffu, 0x80000000u, 8}; uint32_t v = ext[rnd() % 6]; memcpy(&b[rnd() % (b.size() - 4)], &v, 4); } break;
    case 3: if (b.size() > 2) { size_t p = rnd() % b.size(), q = p + rnd() % (b.size() - p); std::vector<uint8_t> s(b.begin() + p, b.begin() + q); b.insert(b.begin() + p, s.begin(), s.end()); } break;
    default: for (int k = 0, n = 1 + rnd() % 64; k < n; k++) b.push_back((uint8_t)rnd());
    }
    return b;
}

int main(int argc, char **argv) {
    if (argc != 8 && argc != 11) { fprintf(stderr, "usage: %s c.zkif i.zkif w.zkif proof.bin label workdir iterations [snark_comm.bin snark_proof.bin nnz]\n", argv[0]); return 2; }
    std::vector<uint8_t> files[3] = {slurp(argv[1]), slurp(argv[2]), slurp(argv[3])}, proof = slurp(argv[4]);
    const std::string label = argv[5], work = argv[6]; const int iters = atoi(argv[7]);
    const std::string tmp[3] = {work + "/m.zkif", work + "/m.inp.zkif", work + "/m.wit.zkif"};
    otti_r1cs *r = zkif_load_impl(argv[1], argv[2], argv[3]);
    auto I = instance_new(r->num_cons, r->num_vars, r->num_inputs, r->A, r->nA, r->B, r->nB, r->C, r->nC);
    auto G = gens_new(r->num_cons, r->num_vars, r->num_inputs);
    std::vector<Fr> inputs(r->ninputs);
    for (size_t i = 0; i < r->ninputs; i++) if (!fr_from_bytes(inputs[i], r->inputs32 + 32 * i)) return 3;
    if (nizk_verify(*I, inputs, *G, label.data(), label.size(), proof.data(), proof.size()) != OTTI_OK) { fprintf(stderr, "the unmodified proof does not verify\n"); return 3; }
    int loaded = 0, refused = 0, accepted = 0, rejected = 0;
    for (int it = 0; it < iters; it++) {
        const int which = (int)(rnd() % 3);
        for (int k = 0; k < 3; k++) spit(tmp[k], k == which ? mutate(files[k]) : files[k]);
        try {
            otti_r1cs *m = zkif_load_impl(tmp[0].c_str(), tmp[1].c_str(), it % 7 == 0 ? nullptr : tmp[2].c_str());
            try { auto J = instance_new(m->num_cons, m->num_vars, m->num_inputs, m->A, m->nA, m->B, m->nB, m->C, m->nC); loaded++; } catch (const Error &) { refused++; }
            otti_r1cs_free(m);
        } catch (const Error &) { refused++; }
        std::vector<uint8_t> mp = mutate(proof);
        int rc;
        try { rc = nizk_verify(*I, inputs, *G, label.data(), label.size(), mp.data(), mp.size()); } catch (const Error &e) { rc = e.code; }
        if (rc == OTTI_OK) { accepted++; if (mp != proof) { fprintf(stderr, "a modified proof was accepted\n"); return 4; } } else rejected++;
    }
    int s_accepted = 0, s_rejected = 0;
    if (argc == 11) {
        const std::vector<uint8_t> cb = slurp(argv[8]), sp = slurp(argv[9]);
        auto SG = snark_gens_new(r->num_cons, r->num_vars, r->num_inputs, (size_t)atoll(argv[10]));
        auto CC = CompComm::parse(cb.data(), cb.size());
        if (snark_verify(*CC, inputs, *SG, label.data(), label.size(), sp.data(), sp.size()) != OTTI_OK) { fprintf(stderr, "the unmodified SNARK proof does not verify\n"); return 3; }
        for (int it = 0; it < iters / 4 + 1; it++) {
            std::vector<uint8_t> mp = mutate(sp);
            int rc;
            try { rc = snark_verify(*CC, inputs, *SG, label.data(), label.size(), mp.data(), mp.size()); } catch (const Error &e) { rc = e.code; }
            if (rc == OTTI_OK) { s_accepted++; if (mp != sp) { fprintf(stderr, "a modified SNARK proof was accepted\n"); return 4; } } else s_rejected++;
        }
    }
    otti_r1cs_free(r);
    printf("sanitized run: zkif %d loaded / %d refused; proofs %d accepted / %d rejected; SNARK proofs %d accepted / %d rejected\n", loaded, refused, accepted, rejected, s_accepted, s_rejected);
    return 0;
}
