// Device-side interface of SNARK mode's kernels (k_snark.hip) for snark_prover.cpp.
#pragma once
#include "device.h"

namespace otti {

constexpr int kMaxInst = 20;                                  // 12 product circuits + 6 dot-product halves in the largest batch
struct AbcList { const Fr *A[kMaxInst], *B[kMaxInst], *C[kMaxInst]; int n; };
struct PtrList { Fr *p[64]; int n; };
struct LayerList { const Fr *in_left[16], *in_right[16]; Fr *out_left[16], *out_right[16]; int n; };

// one round of the layered sum-check over a batch of instances: A, B of every instance are folded in place; C[y] == nullptr means the
// instance's third table is the shared eq table, which is never stored (EqSrc, as in phase one of the R1CS proof)
struct PcList { Fr *A[kMaxInst], *B[kMaxInst], *C[kMaxInst]; int n; };
// Sums S_t (t = 0, 2, 3) of instance y land in c.h_results[slot + 3 y ..] once c.wait_ticket(ticket) returns:
//   C[y] given: S_t = sum_i (A_t B_t C_t)[i];  shared eq: S_t = sum_i E[i] (A_t B_t)[i]  (the host applies the bound variable's factor)
unsigned long long dev_pc_eval(DevCtx &c, const PcList &L, size_t len, const EqSrc &E, int slot);
// r == nullptr: an ARMED launch (device.h) — the fold challenge is the next value the host publishes with c.go()
unsigned long long dev_pc_fold_eval(DevCtx &c, const PcList &L, size_t len, const Fr *r, const EqSrc &E, int slot);   // len >= 4: folds to len / 2 first
// the tables (folded by r first when fold is set) go to pinned host memory: table t of instance y at c.h_results[slot + (3 y + t) * len_out ..)
unsigned long long dev_pc_export(DevCtx &c, const PcList &L, size_t len, bool fold, const Fr *r, int slot);

// ---- the persistent tail of a layer's batched sum-check (k_pc_tail): ONE launch plays every round from the point where the tables of
// an instance fit the LDS of W workgroups (kTailCap elements per table and workgroup) down to the host-played tail.  Workgroup (w, y)
// holds the elements i = w (mod W) of instance y's tables (bound_poly_var_top pairs i with i + len/2: always in the same residue
// class) in LDS for the whole launch; per round it mails its three partial sums to its own TailMail line in pinned host memory, waits
// for the challenge (an armed fetch per round: consecutive go() numbers), folds in LDS.  The eq table of a product-circuit instance
// is materialised in LDS here (a real third table, folded like the others; the host multiplies by the factor accumulated before).
// No launch, no HBM round trip, no inter-workgroup hand-off per round: a round costs the host's hash plus ~3 us of arithmetic.
constexpr int kTailCap = 1024, kTailThreads = 1024, kTailMaxGroups = 160;
// 128 bytes per workgroup = one line, written by ONE store instruction (eight lanes x 16 bytes, system scope) and not followed by a fence: the three
// sums, then (number, tag).  A release fence per mail is what a round of 128-144 workgroups waited for: all mails in after 8.3 us with
// "96 bytes, __threadfence_system, number", 3.35 us with the whole line in one instruction (tools/pollprobe.hip, profiles/r4_pollprobe_mail.txt).
// Nothing orders the two 64-byte halves of the line on their way: tag = go_tag(seq, s, 3) (device.h) lets the host tell a line that is not whole yet.
// (Partial stores without a fence are no alternative: seven 16-byte stores nobody waited for reached the host 13 us later.)
struct TailMail { Fr s[3]; unsigned long long seq, tag, pad[2]; };
static_assert(sizeof(TailMail) == 128, "one mail = one 128-byte line");
struct TailPlan { int W = 0; size_t k0 = 0; };                         // workgroups per instance; first round played by the tail (== ndev: no tail)
// rounds played = log2(len0 / t_out) (>= 1).  Round j's partial sums of workgroup (w, y) arrive in c.h_tail[y * W + w] with seq = first_seq + j;
// after the last fold the tables (t_out elements each) go to c.h_results[slot + (3 y + t) * t_out ..) and every workgroup posts first_seq + rounds.
// fold_r: the source tables hold 2 * len0 elements and are folded by *fold_r on load (nullptr: they hold len0 elements as they are).
// Consumes `rounds` go() values.  Returns first_seq.
unsigned long long dev_pc_tail(DevCtx &c, const PcList &L, int W, size_t len0, size_t t_out, const Fr *fold_r, const EqSrc &E, int slot);

void dev_gather(DevCtx &c, const Fr *table, const uint32_t *idx, Fr *out, size_t n);
void dev_u32_to_fr(DevCtx &c, const uint32_t *in, Fr *out, size_t n);        // out[i] = in[i] as a field element (Montgomery form)
void dev_hash_mem(DevCtx &c, const Fr *eval_table, const Fr *audit_ts, Fr *out_init, Fr *out_audit, size_t M, const Fr &r, const Fr &gamma, int G = 1, int rk = 0);   // G ranks: this rank's residue class (M / G elements) of the M-element vectors
void dev_hash_ops(DevCtx &c, const Fr *addr_f, const Fr *deref, const Fr *read_ts, Fr *out_read, Fr *out_write, size_t N, const Fr &r, const Fr &gamma, int G = 1, int rk = 0);
void dev_prod_layer(DevCtx &c, const LayerList &L, size_t q);
// partials: >= 3 * kMaxBlocks elements of scratch.  Results arrive in c.h_results[slot ..] once the stream has been synchronised.
void dev_pick0(DevCtx &c, const PtrList &L, int slot);
void dev_dot_many(DevCtx &c, const Fr *E, const PtrList &L, size_t n, Fr *partials, int slot);
void dev_sum3(DevCtx &c, const AbcList &L, size_t n, Fr *partials, int slot);

}  // namespace otti
