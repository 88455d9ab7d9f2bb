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

void dev_gather(DevCtx &c, const Fr *table, const uint32_t *idx, Fr *out, size_t n);
void dev_u32_to_fr(DevCtx &c, const uint32_t *in, Fr *out, size_t n);        // out[i] = in[i] as a field element (Montgomery form)
void dev_hash_mem(DevCtx &c, const Fr *eval_table, const Fr *audit_ts, Fr *out_init, Fr *out_audit, size_t M, const Fr &r, const Fr &gamma);
void dev_hash_ops(DevCtx &c, const Fr *addr_f, const Fr *deref, const Fr *read_ts, Fr *out_read, Fr *out_write, size_t N, const Fr &r, const Fr &gamma);
void dev_prod_layer(DevCtx &c, const LayerList &L, size_t q);
// partials: >= 3 * kMaxBlocks elements of scratch.  Results arrive in c.h_results[slot ..] once the stream has been synchronised.
void dev_pick0(DevCtx &c, const PtrList &L, int slot);
void dev_dot_many(DevCtx &c, const Fr *E, const PtrList &L, size_t n, Fr *partials, int slot);
void dev_sum3(DevCtx &c, const AbcList &L, size_t n, Fr *partials, int slot);

}  // namespace otti
