// Where do the microseconds between the host's go() and a workgroup's answer go?  Pieces of the armed-launch / persistent-tail round trip, each
// measured alone (1000 samples, medians; device times from s_memrealtime, 100 MHz):
//   (1) a system-scope load of a pinned host line as the device sees it (one 16-byte load, and the batch of three armed_fetch issues);
//   (2) one hop through an HBM word between two workgroups (agent-scope store -> agent-scope poll), same XCD and different XCDs, and the same
//       hop followed by a dependent sc1 load of a 32-byte value (what a non-leader workgroup of armed_fetch does);
//   (3) device -> host -> device: the device mails (a) a sequence number alone, (b) 96 bytes + __threadfence_system + the number, into a
//       pinned line; a host thread spinning on it answers at once in another pinned line the device polls — the loop time seen by the device.
//   hipcc -O2 --offload-arch=gfx950 tools/rttprobe.hip -o tools/rttprobe.bin -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %s (%d) at line %d\n", hipGetErrorString(r_), (int)r_, __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kSamples = 1000;
constexpr unsigned long long kGiveUp = 200000000ull;        // 2 s in 10 ns ticks: never hang the box

// the same line read by ONE load instruction of `lanes` lanes x 16 bytes (what the memory pipeline makes of adjacent lanes: one 64-byte request?)
__global__ void k_host_load_wide(const unsigned long long *line, int lanes, unsigned *out) {
    for (int i = 0; i < kSamples; i++) {
        u32x4 a = {0u, 0u, 0u, 0u};
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if ((int)threadIdx.x < lanes) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"((const char *)line + 16 * threadIdx.x) : "memory");
        const unsigned x = __shfl(a[0], 0) ^ __shfl(a[1], lanes - 1);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) out[i] = (unsigned)(t1 - t0) + (x & 0u);
        __builtin_amdgcn_s_sleep(8);
    }
}
__global__ void k_host_load(const unsigned long long *line, unsigned *out1, unsigned *out3) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < kSamples; i++) {
        u32x4 a, b, c;
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(line) : "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        out1[i] = (unsigned)(t1 - t0) + (a[0] & 0u);
        t0 = __builtin_amdgcn_s_memrealtime();
        asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:48 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(line) : "memory");
        t1 = __builtin_amdgcn_s_memrealtime();
        out3[i] = (unsigned)(t1 - t0) + ((a[0] ^ b[0] ^ c[0]) & 0u);
        __builtin_amdgcn_s_sleep(8);
    }
}
// workgroups `a` and `b` of the grid play ping-pong through two HBM words; with_value: the receiver then loads 32 bytes next to the word (sc1)
__global__ void k_hop(unsigned long long *ping, unsigned long long *pong, uint32_t *value, int a, int b, int with_value, unsigned *out, unsigned *xcc) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x == a ? 0 : 1] = id & 0xf;
    const unsigned long long start = __builtin_amdgcn_s_memrealtime();
    uint32_t sink = 0;
    for (unsigned long long k = 1; k <= kSamples; k++) {
        if ((int)blockIdx.x == a) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            if (with_value) { for (int i = 0; i < 8; i++) __builtin_nontemporal_store((uint32_t)k + i, &value[i]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_store(ping, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(pong, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < k) { if (__builtin_amdgcn_s_memrealtime() - start > kGiveUp) return; __builtin_amdgcn_s_sleep(1); }
            out[k - 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
        } else {
            while (__hip_atomic_load(ping, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) { if (__builtin_amdgcn_s_memrealtime() - start > kGiveUp) return; __builtin_amdgcn_s_sleep(1); }
            if (with_value) {
                u32x4 v0, v1;
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(value) : "memory");
                sink += v0[0] + v1[3];
            }
            __hip_atomic_store(pong, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (sink == 0x12345678u) out[0] = 0;
}
// the same hop with the value travelling WITH the word: the sender stores 32 bytes and then (number, check) as three 16-byte sc1 stores without waiting;
// the receiver polls with one batch of three 16-byte sc1 loads until number and check fit the value
__global__ void k_hop_batch(uint32_t *ping, uint32_t *pong, int a, int b, unsigned *out) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    const unsigned long long start = __builtin_amdgcn_s_memrealtime();
    const bool sender = (int)blockIdx.x == a;
    for (uint32_t k = 1; k <= kSamples; k++) {
        uint32_t *mine = sender ? ping : pong, *theirs = sender ? pong : ping;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        auto post = [&](uint32_t *box) {
            u32x4 v0, v1, h; for (int i = 0; i < 4; i++) { v0[i] = k * 7 + i; v1[i] = k * 11 + i; } h[0] = k; h[1] = 0; h[2] = k * 7 + k * 11 + 3; h[3] = 0;
            asm volatile("global_store_dwordx4 %0, %1, off offset:32 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:48 sc1\n\tglobal_store_dwordx4 %0, %3, off sc1" :: "v"(box), "v"(v0), "v"(v1), "v"(h) : "memory");
        };
        auto await = [&](const uint32_t *box) {
            for (;;) {
                u32x4 h, v0, v1;
                asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:32 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:48 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(h), "=&v"(v0), "=&v"(v1) : "v"(box) : "memory");
                if (h[0] == k && h[2] == v0[0] + v1[3]) return true;
                if (__builtin_amdgcn_s_memrealtime() - start > kGiveUp) return false;
                __builtin_amdgcn_s_sleep(1);
            }
        };
        if (sender) { post(mine); if (!await(theirs)) return; out[k - 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0); }
        else { if (!await(theirs)) return; post(mine); }
    }
}
struct alignas(128) Mail { unsigned long long s[12]; unsigned long long seq; unsigned long long pad[3]; };
__global__ void k_loop(Mail *mail, const unsigned long long *door, int with_data, unsigned *out) {
    if (threadIdx.x != 0) return;
    const unsigned long long start = __builtin_amdgcn_s_memrealtime();
    for (unsigned long long k = 1; k <= kSamples; k++) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (with_data) { for (int i = 0; i < 12; i++) mail->s[i] = k + i; __threadfence_system(); }
        __hip_atomic_store(&mail->seq, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        for (;;) {
            u32x4 a;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(door) : "memory");
            if (((unsigned long long)a[0] | ((unsigned long long)a[1] << 32)) >= k) break;
            if (__builtin_amdgcn_s_memrealtime() - start > kGiveUp) return;
            __builtin_amdgcn_s_sleep(1);
        }
        out[k - 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
        __builtin_amdgcn_s_sleep(32);
    }
}
static void report(const char *what, unsigned *v) {
    std::vector<unsigned> s(v, v + kSamples); std::sort(s.begin(), s.end());
    printf("%-118s median %6.2f us  p10 %6.2f  p90 %6.2f\n", what, 0.01 * s[kSamples / 2], 0.01 * s[kSamples / 10], 0.01 * s[kSamples * 9 / 10]);
}
int main() {
    unsigned long long *line_h, *line_d; Mail *mail_h, *mail_d; unsigned *out_h, *out_d, *out2_h, *out2_d, *xcc_h, *xcc_d;
    CHECK(hipHostMalloc((void **)&line_h, 256, hipHostMallocCoherent | hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&line_d, line_h, 0));
    CHECK(hipHostMalloc((void **)&mail_h, sizeof(Mail), hipHostMallocCoherent | hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&mail_d, mail_h, 0));
    CHECK(hipHostMalloc((void **)&out_h, 4 * kSamples, hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&out_d, out_h, 0));
    CHECK(hipHostMalloc((void **)&out2_h, 4 * kSamples, hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&out2_d, out2_h, 0));
    CHECK(hipHostMalloc((void **)&xcc_h, 64, hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&xcc_d, xcc_h, 0));
    memset(line_h, 0, 256); memset(mail_h, 0, sizeof(Mail));
    // (1)
    hipLaunchKernelGGL(k_host_load, 1, 64, 0, 0, (const unsigned long long *)line_d, out_d, out2_d); CHECK(hipDeviceSynchronize());
    report("(1) system-scope load of a pinned host line, 16 bytes", out_h); report("(1) the same, three 16-byte loads of the line in one batch (armed_fetch's poll)", out2_h);
    for (int lanes : {1, 2, 4, 8, 10}) {
        hipLaunchKernelGGL(k_host_load_wide, 1, 64, 0, 0, (const unsigned long long *)line_d, lanes, out_d); CHECK(hipDeviceSynchronize());
        char what[200]; snprintf(what, sizeof what, "(1) the line read by ONE load instruction, %d lane(s) x 16 bytes", lanes);
        report(what, out_h);
    }
    // (2)
    unsigned long long *words; uint32_t *value; CHECK(hipMalloc((void **)&words, 512)); CHECK(hipMalloc((void **)&value, 256));
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
    for (int with_value = 0; with_value < 2; with_value++)
        for (int p = 0; p < 3; p++) {
            CHECK(hipMemset(words, 0, 512)); memset(out_h, 0, 4 * kSamples);
            hipLaunchKernelGGL(k_hop, 16, 64, 0, 0, words, words + 16, value, pairs[p][0], pairs[p][1], with_value, out_d, xcc_d); CHECK(hipDeviceSynchronize());
            char what[200]; snprintf(what, sizeof what, "(2) there and back through HBM words, workgroups %d and %d (XCC %u and %u)%s", pairs[p][0], pairs[p][1], xcc_h[0], xcc_h[1], with_value ? ", the receiver also loads 32 bytes (sc1) once the word is in" : "");
            report(what, out_h);
        }
    for (int p = 0; p < 3; p++) {
        CHECK(hipMemset(words, 0, 512)); memset(out_h, 0, 4 * kSamples);
        hipLaunchKernelGGL(k_hop_batch, 16, 64, 0, 0, (uint32_t *)words, (uint32_t *)(words + 32), pairs[p][0], pairs[p][1], out_d); CHECK(hipDeviceSynchronize());
        char what[200]; snprintf(what, sizeof what, "(2) there and back, workgroups %d and %d: 32 bytes + (number, check) as three 16-byte sc1 stores, polled as one batch of loads", pairs[p][0], pairs[p][1]);
        report(what, out_h);
    }
    // (3)
    for (int with_data = 0; with_data < 2; with_data++) {
        memset(line_h, 0, 256); memset(mail_h, 0, sizeof(Mail)); memset(out_h, 0, 4 * kSamples);
        std::atomic<bool> stop{false};
        std::thread echo([&] { unsigned long long seen = 0; while (!stop.load(std::memory_order_relaxed)) { const unsigned long long s = __atomic_load_n(&mail_h->seq, __ATOMIC_ACQUIRE); if (s > seen) { seen = s; __atomic_store_n(line_h, s, __ATOMIC_RELEASE); } } });
        hipLaunchKernelGGL(k_loop, 1, 64, 0, 0, mail_d, (const unsigned long long *)line_d, with_data, out_d); CHECK(hipDeviceSynchronize());
        stop.store(true); echo.join();
        report(with_data ? "(3) device mails 96 bytes + __threadfence_system + number, host answers at once, device polls the answer" : "(3) device mails a number, host answers at once, device polls the answer", out_h);
    }
    return 0;
}
