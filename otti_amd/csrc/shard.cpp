// See shard.h.  Layout of the segment: a 4 KiB header (magic, world size, then one 64-byte line per rank holding that rank's epoch
// counter), followed by 2 * world payload slots of kSlotBytes.  Epoch e uses the slots of parity e & 1: a rank can only reach epoch
// e + 2 (and overwrite parity-e slots) after every rank has posted epoch e + 1, which each of them does only after it has finished
// reading epoch e — so two slot sets are enough and no second barrier is needed.
#include "shard.h"
#include "device.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <vector>
#include <errno.h>
#include <fcntl.h>
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace otti {

namespace {
constexpr uint64_t kMagic = 0x6f7474692d736864ULL;   // "otti-shd"
constexpr size_t kHeader = 4096;
struct Header { std::atomic<uint64_t> magic; uint64_t world; };
static_assert(sizeof(std::atomic<uint64_t>) == 8, "lock-free 64-bit atomics expected");

double timeout_s() { const char *e = getenv("OTTI_SHARD_TIMEOUT_S"); double t = e ? atof(e) : 120.0; return t > 0 ? t : 120.0; }

struct Deadline {
    std::chrono::steady_clock::time_point end = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(timeout_s()));
    unsigned spins = 0;
    void pause(const char *what) {
        _mm_pause();
        if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() > end) throw Error(OTTI_ERR_INTERNAL, std::string("shard exchange timed out: ") + what);
    }
};
std::atomic<uint64_t> *seq_of(uint8_t *base, int r) { return reinterpret_cast<std::atomic<uint64_t> *>(base + 64 * (size_t)(r + 1)); }
}  // namespace

// ---- RCCL transport.  librccl is loaded at run time (the library carries no link-time dependency on it; a process that already
// holds PyTorch's copy gets that one), the communicator is bootstrapped through the mailbox (rank 0's ncclUniqueId).
struct ShardComm::Rccl {
    void *lib = nullptr; ncclComm_t comm = nullptr; hipStream_t stream = nullptr;
    uint64_t *h_lanes = nullptr, *d_lanes = nullptr; size_t cap = 0;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    void check(ncclResult_t r, const char *what) { if (r != ncclSuccess) throw Error(OTTI_ERR_NO_DEVICE, std::string("RCCL: ") + what + ": " + (GetErrorString ? GetErrorString(r) : "error")); }
    explicit Rccl(ShardComm &c) {
        DevCtx::get();                                                   // selects this rank's device for the calling thread (OTTI_DEVICE / LOCAL_RANK)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) throw Error(OTTI_ERR_NO_DEVICE, "OTTI_SHARD_TRANSPORT=rccl: librccl.so not found");
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) throw Error(OTTI_ERR_NO_DEVICE, std::string("librccl lacks ") + n); return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId"); CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        ncclUniqueId id; memset(&id, 0, sizeof id);
        if (c.rank() == 0) check(GetUniqueId(&id), "ncclGetUniqueId");
        std::vector<ncclUniqueId> ids((size_t)c.world());
        c.allgather(&id, sizeof id, ids.data());
        check(CommInitRank(&comm, c.world(), ids[0], c.rank()), "ncclCommInitRank");
        OTTI_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        reserve(8 * 64);
    }
    void reserve(size_t lanes) {
        if (lanes <= cap) return;
        if (h_lanes) (void)hipHostFree(h_lanes);
        if (d_lanes) (void)hipFree(d_lanes);
        OTTI_HIP(hipHostMalloc((void **)&h_lanes, lanes * 8, hipHostMallocDefault)); OTTI_HIP(hipMalloc((void **)&d_lanes, lanes * 8)); cap = lanes;
    }
    void allreduce(Fr *v, size_t n) {
        reserve(8 * n);
        for (size_t i = 0; i < n; i++) for (int k = 0; k < 8; k++) h_lanes[8 * i + k] = v[i].v[k];
        OTTI_HIP(hipMemcpyAsync(d_lanes, h_lanes, 64 * n, hipMemcpyHostToDevice, stream));
        check(AllReduce(d_lanes, d_lanes, 8 * n, ncclUint64, ncclSum, comm, stream), "ncclAllReduce");
        OTTI_HIP(hipMemcpyAsync(h_lanes, d_lanes, 64 * n, hipMemcpyDeviceToHost, stream));
        OTTI_HIP(hipStreamSynchronize(stream));
        lanes_to_fr(h_lanes, n, v);
    }
    void allreduce_device(const Fr *d_src, const Fr *factor, Fr *v, size_t n) {
        reserve(8 * n);
        dev_fr_to_lanes(stream, d_src, factor, (unsigned long long *)d_lanes, n);
        check(AllReduce(d_lanes, d_lanes, 8 * n, ncclUint64, ncclSum, comm, stream), "ncclAllReduce");
        OTTI_HIP(hipMemcpyAsync(h_lanes, d_lanes, 64 * n, hipMemcpyDeviceToHost, stream));
        OTTI_HIP(hipStreamSynchronize(stream));
        lanes_to_fr(h_lanes, n, v);
    }
    ~Rccl() {
        if (comm && CommDestroy) (void)CommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
        if (h_lanes) (void)hipHostFree(h_lanes);
        if (d_lanes) (void)hipFree(d_lanes);
    }
};

void lanes_to_fr(const uint64_t *lanes, size_t n, Fr *out) {
    for (size_t i = 0; i < n; i++) {
        uint32_t w[9]; unsigned __int128 c = 0;
        for (int k = 0; k < 8; k++) { c += lanes[8 * i + k]; w[k] = (uint32_t)c; c >>= 32; }
        w[8] = (uint32_t)c;
        // value = lo (256 bits, any value) + hi * 2^256 with hi < 2^32: reduce both through Montgomery products with R^2
        // (x * R^2 / R = x * R, then * 1 / R drops back to x mod l)
        Fr lo, hi = fr_zero(); for (int k = 0; k < 8; k++) lo.v[k] = w[k]; hi.v[0] = w[8];
        Fr one_raw = fr_zero(); one_raw.v[0] = 1;
        out[i] = fr_add(fr_mul(fr_mul(lo, fr_R2()), one_raw), fr_mul(hi, fr_R2()));
    }
}

ShardComm::ShardComm(const std::string &name, int rank, int world) : name_(name), rank_(rank), world_(world) {
    if (world < 1 || world > kMaxWorld || (world & (world - 1)) || rank < 0 || rank >= world) throw Error(OTTI_ERR_BAD_ARG, "shard: world must be a power of two <= 64 and 0 <= rank < world");
    if (name.empty() || name.find('/') != std::string::npos) throw Error(OTTI_ERR_BAD_ARG, "shard: segment name must be non-empty and contain no '/'");
    if (64 * (size_t)(world + 1) > kHeader) throw Error(OTTI_ERR_BAD_ARG, "shard: world too large");
    bytes_ = kHeader + 2 * (size_t)world * kSlotBytes;
    const std::string path = "/" + name;
    Deadline dl;
    if (rank == 0) {
        shm_unlink(path.c_str());                                       // a leftover of a crashed run with the same name
        fd_ = shm_open(path.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd_ < 0) throw Error(OTTI_ERR_IO, "shard: shm_open(create) failed: " + std::string(strerror(errno)));
        if (ftruncate(fd_, (off_t)bytes_) != 0) { int e = errno; close(fd_); shm_unlink(path.c_str()); throw Error(OTTI_ERR_IO, "shard: ftruncate failed: " + std::string(strerror(e))); }
    } else {
        for (;;) {
            fd_ = shm_open(path.c_str(), O_RDWR, 0600);
            if (fd_ >= 0) { struct stat st; if (fstat(fd_, &st) == 0 && (size_t)st.st_size == bytes_) break; close(fd_); fd_ = -1; }
            dl.pause("waiting for rank 0 to create the segment");
        }
    }
    void *p = mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
    if (p == MAP_FAILED) { int e = errno; close(fd_); if (rank == 0) shm_unlink(path.c_str()); throw Error(OTTI_ERR_IO, "shard: mmap failed: " + std::string(strerror(e))); }
    base_ = (uint8_t *)p;
    Header *h = reinterpret_cast<Header *>(base_);
    if (rank == 0) { h->world = (uint64_t)world; h->magic.store(kMagic, std::memory_order_release); }
    else {
        while (h->magic.load(std::memory_order_acquire) != kMagic) dl.pause("waiting for rank 0 to initialise the segment");
        if (h->world != (uint64_t)world) throw Error(OTTI_ERR_BAD_ARG, "shard: segment was created for a different world size");
    }
    barrier();
    if (rank == 0) shm_unlink(path.c_str());                            // everyone is attached: the name can go, the mapping stays
    const char *tr = getenv("OTTI_SHARD_TRANSPORT");
    if (tr && !strcmp(tr, "rccl")) rccl_ = new Rccl(*this);
    else if (tr && *tr && strcmp(tr, "mailbox")) throw Error(OTTI_ERR_BAD_ARG, "OTTI_SHARD_TRANSPORT must be 'mailbox' or 'rccl'");
}

ShardComm::~ShardComm() {
    delete rccl_;
    if (base_) munmap(base_, bytes_);
    if (fd_ >= 0) close(fd_);
}

uint8_t *ShardComm::slot(int parity, int r) const { return base_ + kHeader + ((size_t)parity * world_ + r) * kSlotBytes; }

void ShardComm::allgather(const void *mine, size_t n, void *out) {
    if (n > kSlotBytes) throw Error(OTTI_ERR_BAD_ARG, "shard: payload larger than a slot");
    const uint64_t e = ++epoch_; const int par = (int)(e & 1);
    memcpy(slot(par, rank_), mine, n);
    seq_of(base_, rank_)->store(e, std::memory_order_release);
    Deadline dl;
    for (int r = 0; r < world_; r++) {
        while (seq_of(base_, r)->load(std::memory_order_acquire) < e) dl.pause("a peer did not post");
        memcpy((uint8_t *)out + (size_t)r * n, slot(par, r), n);
    }
}

void ShardComm::allreduce_fr(Fr *v, size_t n) {
    if (rccl_) { rccl_->allreduce(v, n); return; }                      // also with a world of one: the path is exercised on a one-GPU box
    allreduce_mailbox(v, n);
}
void ShardComm::allreduce_fr_device(const Fr *d_src, const Fr *factor, Fr *v, size_t n) {
    if (rccl_ && d_src) { rccl_->allreduce_device(d_src, factor, v, n); return; }
    allreduce_fr(v, n);                                                  // v: the host copy, already multiplied by the caller
}
void ShardComm::allreduce_mailbox(Fr *v, size_t n) {
    if (world_ == 1) return;
    const size_t per = kSlotBytes / sizeof(Fr);
    std::vector<Fr> all;
    for (size_t off = 0; off < n; off += per) {
        const size_t m = std::min(per, n - off);
        all.resize((size_t)world_ * m);
        allgather(v + off, m * sizeof(Fr), all.data());
        for (size_t i = 0; i < m; i++) {
            Fr s = all[i];
            for (int r = 1; r < world_; r++) s = fr_add(s, all[(size_t)r * m + i]);
            v[off + i] = s;
        }
    }
}

static ShardComm *g_comm = nullptr;
ShardComm *shard_comm() { return g_comm; }
void shard_comm_set(ShardComm *c) { delete g_comm; g_comm = c; }

}  // namespace otti
