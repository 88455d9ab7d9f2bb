// One proof over several MI355X of one node: one process per GPU, every process runs the identical Fiat-Shamir
// transcript, and the only data that crosses GPUs while proving is what the HOST needs anyway before it can hash — the per-round
// partial sums of the sum-checks (96 / 64 bytes), the row commitments (32 bytes per matrix row), the partial L^T Z vector (sqrt(V)
// elements) and the last log2(g) table elements.  All of it is at most a few hundred KiB and latency-bound, and it already sits in
// pinned host memory when it is produced (the kernels mail their results there), so the exchange is a shared-memory mailbox between
// the processes of the node: post, flag, spin — about a microsecond per round, against tens of microseconds for a device collective
// on 96 bytes.  Bulk data (the witness) is replicated before the proof starts, by the caller (bench.py: torch.distributed broadcast
// over RCCL/xGMI).
//
// The reference has no multi-device code at all [REF /root/reference/run.py:52-59: one `spzk` process]; this is new design.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include "field.h"

namespace otti {

class ShardComm {
public:
    // Collective over the `world` processes that pass the same name.  Rank 0 creates /dev/shm/<name>; the others attach.
    ShardComm(const std::string &name, int rank, int world);
    ~ShardComm();
    int rank() const { return rank_; }
    int world() const { return world_; }
    // out[r*n .. (r+1)*n) = rank r's `mine`, for every rank; n <= kSlotBytes
    void allgather(const void *mine, size_t n, void *out);
    // v[i] <- sum over ranks of v[i]  (exact arithmetic in GF(l): the order of summation cannot change the result)
    void allreduce_fr(Fr *v, size_t n);
    // the same for values a kernel has just left in HBM (d_src, complete: the caller has seen the kernel's result), each times *factor when
    // given: with the RCCL transport they are packed into lanes on the device and reduced from there — no host pack, no upload, one
    // download of the result; with the mailbox the host copy in v (already multiplied by the caller) is exchanged as always
    void allreduce_fr_device(const Fr *d_src, const Fr *factor, Fr *v, size_t n);
    void barrier() { uint8_t b = 0, all[64]; allgather(&b, 1, all); }
    static constexpr size_t kSlotBytes = (size_t)1 << 20;
    static constexpr int kMaxWorld = 64;
    // Transport of allreduce_fr.  Default: the mailbox above.  OTTI_SHARD_TRANSPORT=rccl: every element travels as 8 x u32 limbs
    // widened to u64 lanes through ncclAllReduce(ncclUint64, ncclSum) on the GPUs (RCCL has no modular reduce op,
    // but world * 2^32 < 2^64, so a plain integer sum of lanes followed by one normalisation mod l is exact), i.e. over xGMI when
    // the ranks sit on different cards.  Selectable so that the two can be compared on a multi-GPU node; the mailbox wins on
    // latency for payloads of 64-96 bytes.  allgather always uses the mailbox.
    enum Transport { kMailbox = 0, kRccl = 1 };
    Transport transport() const { return rccl_ ? kRccl : kMailbox; }
private:
    uint8_t *slot(int parity, int r) const;
    void allreduce_mailbox(Fr *v, size_t n);
    std::string name_; int rank_, world_; int fd_ = -1; uint8_t *base_ = nullptr; size_t bytes_ = 0; uint64_t epoch_ = 0;
    struct Rccl; Rccl *rccl_ = nullptr;
};
// sum of lane groups back to field elements: each group of 8 u64 lanes holds a sum of up to 2^32 32-bit limbs of Montgomery-form
// values; carries are propagated and the 288-bit result reduced mod l (still Montgomery form: the sum of Montgomery values is the
// Montgomery value of the sum)
void lanes_to_fr(const uint64_t *lanes, size_t n, Fr *out);

// process-wide communicator used by nizk_prove_sharded (set through the C ABI: otti_shard_init / otti_shard_finalize)
ShardComm *shard_comm();
void shard_comm_set(ShardComm *c);

}  // namespace otti
