// Host-only GF(2^255-19) arithmetic in five 51-bit limbs, for the two things the prover's HOST does between two device launches of
// a sum-check or bullet round: compress a point (an inverse square root: ~254 dependent squarings), a fixed-base scalar
// multiplication (32 mixed additions from an 8-bit window table).  These sit on the strictly sequential
// Fiat-Shamir path — about three compressions per round, 41 + 10 rounds at 2^20 — so their latency is proof latency.  The generic
// 4 x u64 schoolbook product of field.h (shared with the device code, which uses other limb forms) costs ~27 ns; the unsaturated
// form needs no carry chain inside the product and has a dedicated squaring.
// Same values as point.h / field.h: every function here is checked against them by otti_host_selftest (tests/test_host.py).
#pragma once
#include "point.h"

namespace otti {

struct Fe { uint64_t v[5]; };
struct PtFe { Fe X, Y, Z, T; };
struct NielsFe { Fe yplusx, yminusx, xy2d; };               // 120 B

Fe fe_from_fp(const Fp &a);
Fp fe_to_fp(const Fe &a);
PtFe ptfe_from(const Pt &p);
Pt ptfe_to(const PtFe &p);
NielsFe nielsfe_from(const Niels &n);
void ptfe_madd(PtFe &p, const NielsFe &q, bool negate);      // p += q (or -= q)
void ptfe_add(PtFe &p, const PtFe &q);                       // p += q (unified extended addition, 9M)
PtFe ptfe_identity();
// the verifier's variable-base work (scalar multiplications of proof points, MSMs over decompressed commitments): doubling, and
// addition of a point kept in "cached" form (Y+X, Y-X, 2Z, 2dT: 8M, negation is a swap)
struct CachedFe { Fe yplusx, yminusx, z2, t2d; };
CachedFe ptfe_cache(const PtFe &p);
void ptfe_dbl(PtFe &p);                                      // p = 2p (4S + 4M)
void ptfe_add_cached(PtFe &p, const CachedFe &q, bool negate);

// RFC 9496 4.3.2 Encode; identical output to pt_encode (point.h)
void pt_encode_fast(uint8_t out[32], const Pt &p);
void pt_encode_fe(uint8_t out[32], const PtFe &p);           // the same from the five-limb form
// RFC 9496 4.3.1 Decode; identical result to pt_decode (point.h); false = DecompressionError
bool pt_decode_fast(Pt &out, const uint8_t b[32]);

// 1 / a in GF(l) (Montgomery form in and out; 0 -> 0), variable time: Bernstein-Yang division steps, 62 at a time on the low words with the
// transition matrix applied to the full numbers afterwards — ~1.5 us against the 8.8 us of the exponentiation a^(l-2) (fr_inv, field.h).  The
// log-size dot-product proofs invert one challenge per round, on the host, between two device launches: 10 rounds per NIZK proof, 55 per
// SNARK proof.  The result is checked (one product) and the exponentiation answers if the check ever failed.
Fr fr_inv_fast(const Fr &a);
bool fr_inv_fast_try(const Fr &a, Fr &out);                 // the same without the fallback (selftest): false if the result failed its check

}  // namespace otti
