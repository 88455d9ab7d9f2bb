// The last rounds of a batched cubic sum-check, played by the host (SNARK mode: every layer of the product circuits ends with 4-6 rounds
// over tables of 16-64 elements per instance, 41 layers per proof: a device launch per such round costs more than its arithmetic).
// Restates SumcheckInstanceProof::prove_cubic_batched of upstream libspartan's sumcheck.rs for tables held in host memory [RECALL;
// /root/reference/Spartan is an empty submodule]; CPU restatement: the oracle's snark.c (test infrastructure).
//
// Two implementations behind one interface, same field elements out of both:
//   * scalar: the generic 4 x u64 Montgomery code of field.h, instances spread over the prover's helper threads;
//   * AVX-512 IFMA (where the CPU has it and OTTI_HOST_FR8 is not 0): eight products in GF(l) at a time — five 52-bit limbs per element,
//     one element per 64-bit lane, Montgomery reduction by 2^260 (vpmadd52luq / vpmadd52huq) — with four instances x two consecutive table
//     elements per vector, so that a round's (i, i + len/2) pairs are whole vectors.  ~4.5 ns per product against ~27.
#pragma once
#include <memory>
#include <vector>
#include "field.h"

namespace otti {

class HostTail {
public:
    // np product instances (tables A_k, B_k and the eq table E shared by all of them), then nd triples (A_k, B_k, C_k): k = np .. np + nd - 1.
    // Every table has T elements (a power of two, >= 2).  C[k] is read for k >= np only; E may be null when np == 0.
    // coeff[k]: the batching coefficient of instance k.  threads: how many threads a round may use (the caller's SpinPool).
    static std::unique_ptr<HostTail> make(int np, int nd, size_t T, const Fr *const *A, const Fr *const *B, const Fr *const *C, const Fr *E, const Fr *coeff, int threads, bool force_scalar = false);
    virtual ~HostTail() {}
    // this round's combined evaluations: out[t] = sum_k coeff_k * sum_i (A_k B_k C_k)(point t of pair i), points 0, 2, 3
    virtual void sums(Fr out[3]) = 0;
    virtual void fold(const Fr &r) = 0;                      // bound_poly_var_top of every table
    virtual size_t len() const = 0;
    virtual void last(int k, Fr out[3]) const = 0;           // A_k[0], B_k[0], C_k[0] once len() == 1 (the third is E[0] for a product instance)
    virtual const char *kind() const = 0;
};
bool host_fr8_available();
// out0[t] = sum_{k < n0} w_k s[k][t], out1[t] = sum_{n0 <= k < n} w_k s[k][t] for t = 0, 1, 2 — a device round's per-instance sums (3 each) times the
// batching coefficients, product instances and triples apart; n <= 24.  Eight products at a time where the CPU has the instructions.
void weighted_sums3(const Fr *s, const Fr *w, int n0, int n, Fr out0[3], Fr out1[3]);
// selftest hook (otti_host_selftest): both implementations on the same random tables; throws on the first difference
void hosttail_selftest(uint32_t seed);
// measurement aid (otti_host_tail_bench): microseconds per layer (every round's sums + fold) on random tables; out[0]: the vector form (0 without the instructions), out[1]: scalar
void hosttail_bench(int np, int nd, size_t T, int threads, int reps, double out[2]);

}  // namespace otti
