// How should the P workgroups of a persistent launch learn a value the host publishes?  (a) every workgroup's lane 0 polls the pinned host
// line itself (P pollers on the PCIe link), (b) workgroup 0 polls the host line and republishes in HBM, the others poll HBM (what armed_fetch
// does).  Each workgroup then echoes into a pinned host line of its own (as the sum-check tail mails its partial sums); the host measures
// the time from its store to the LAST echo, 2000 rounds, P = 1, 16, 64, 144.
// `pollprobe.bin vram`: the host's line in fine-grained GPU memory instead (the CPU stores through the PCIe BAR, every workgroup polls local memory, no relay):
// the stores arrive, but the polls see them after anything between 2 us and 10 ms (profiles/r4_pollprobe_vram.txt) — not usable.
//   hipcc -O2 --offload-arch=gfx950 tools/pollprobe.hip -o tools/pollprobe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <chrono>
#include <algorithm>
#include <vector>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %s (%d) at line %d\n", hipGetErrorString(r_), (int)r_, __LINE__); return 1; } } while (0)

struct alignas(128) Line { unsigned long long v; unsigned long long pad[15]; };

__global__ __launch_bounds__(1024) void k_poll(const unsigned long long *door, unsigned long long *relay, Line *echo, unsigned long long rounds, int mode, int sleep, int fenced) {
    __shared__ unsigned long long s_seen;
    extern __shared__ unsigned char big_lds[];               // (the tail's workgroups hold 96 KB each: one per CU)
    if (threadIdx.x == 1023 && rounds == 0) big_lds[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned long long want = 1; want <= rounds; want++) {
        if (threadIdx.x == 0) {
            unsigned long long s = 0;
            if (mode == 0 || blockIdx.x == 0) {
                for (;;) {
                    s = __hip_atomic_load(door, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (s >= want) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 1000000000ull) { s = ~0ull; break; }      // 10 s: never hang the box
                    if (sleep) __builtin_amdgcn_s_sleep(1);
                }
                if (mode == 1) __hip_atomic_store(relay, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (;;) {
                    s = __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (s >= want) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 1000000000ull) { s = ~0ull; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            s_seen = s;
        }
        __syncthreads();
        if (s_seen == ~0ull) return;
        if (fenced == 2) {
            // the whole 128-byte line in ONE store instruction (eight lanes x 16 bytes, system scope), no fence: number in the first 16 bytes, a check word in the last
            if (threadIdx.x < 8) {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                u32x4 q; q[0] = (unsigned)want; q[1] = (unsigned)(want >> 32); q[2] = (unsigned)want * 7u + threadIdx.x; q[3] = threadIdx.x == 7 ? (unsigned)want * 3u : 0u;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"((char *)&echo[blockIdx.x] + 16 * threadIdx.x), "v"(q) : "memory");
            }
        } else if (threadIdx.x == 0) {
            if (fenced) { for (int i = 0; i < 12; i++) echo[blockIdx.x].pad[i] = want + i; __threadfence_system(); }      // 96 bytes of data, release fence, number: a round's mail
            __hip_atomic_store(&echo[blockIdx.x].v, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
}
static int measure(int P, int mode, int sleep, int fenced = 0, int threads = 256, int lds = 0, int vram_door = 0) {
    unsigned long long *door_h, *door_d, *relay; Line *echo_h, *echo_d;
    if (vram_door) {   // the line the host publishes in lives in the GPU's own memory (fine-grained; the CPU stores through the PCIe BAR): every workgroup polls LOCAL memory
        CHECK(hipExtMallocWithFlags((void **)&door_d, 128, hipDeviceMallocFinegrained)); CHECK(hipMemset(door_d, 0, 128)); CHECK(hipDeviceSynchronize());
        door_h = door_d;
    } else { CHECK(hipHostMalloc((void **)&door_h, 128, hipHostMallocCoherent | hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&door_d, door_h, 0)); }
    CHECK(hipHostMalloc((void **)&echo_h, sizeof(Line) * P, hipHostMallocCoherent | hipHostMallocMapped)); CHECK(hipHostGetDevicePointer((void **)&echo_d, echo_h, 0));
    CHECK(hipMalloc((void **)&relay, 128)); CHECK(hipMemset(relay, 0, 128));
    memset(echo_h, 0, sizeof(Line) * P); if (!vram_door) *door_h = 0;
    const unsigned long long rounds = 2000;
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (lds) CHECK(hipFuncSetAttribute((const void *)k_poll, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k_poll, P, threads, lds, st, (const unsigned long long *)door_d, relay, echo_d, rounds, mode, sleep, fenced);
    usleep(20000);
    std::vector<double> last, first; last.reserve(rounds); first.reserve(rounds);
    for (unsigned long long r = 1; r <= rounds; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(door_h, r, __ATOMIC_RELEASE);
        double t_first = -1;
        for (int i = 0; i < P; i++) {
            while (__atomic_load_n(&echo_h[i].v, __ATOMIC_ACQUIRE) < r || (fenced == 2 && (unsigned)(__atomic_load_n(&echo_h[i].pad[14], __ATOMIC_ACQUIRE) >> 32) != (unsigned)r * 3u))
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { printf("P=%d mode %d: echo %d never came (round %llu)\n", P, mode, i, r); __atomic_store_n(door_h, rounds + 1, __ATOMIC_RELEASE); (void)hipStreamSynchronize(st); return 2; }
            if (i == 0) t_first = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        last.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); first.push_back(t_first);
        for (volatile int spin = 0; spin < 2000; spin++) {}                  // a few microseconds between rounds, as the host's hashing leaves
    }
    CHECK(hipStreamSynchronize(st));
    std::sort(last.begin(), last.end()); std::sort(first.begin(), first.end());
    printf("P=%3d %s%s%-58s all echoes in: median %.2f us, p10 %.2f, p90 %.2f, p99 %.2f  (workgroup 0's: median %.2f)\n", P,
           vram_door ? "[the host's line in GPU memory, written through the BAR] " : "", fenced == 2 ? (threads == 1024 ? "[1024 threads, 96 KB LDS, mail = the 128-byte line in one store instruction, no fence] " : "[mail = the 128-byte line in one store instruction, no fence] ") : fenced ? (threads == 1024 ? "[1024 threads, 96 KB LDS, mail = 96 B + fence + number] " : "[mail = 96 B + fence + number] ") : "", mode == 0 ? (sleep ? "every workgroup polls the host line (s_sleep 1 between polls)" : "every workgroup polls the host line (no sleep)") : "workgroup 0 polls the host line, republishes in HBM",
           last[last.size() / 2], last[last.size() / 10], last[last.size() * 9 / 10], last[last.size() * 99 / 100], first[first.size() / 2]);
    (void)fenced;
    if (vram_door) (void)hipFree(door_d); else (void)hipHostFree(door_h);
    (void)hipHostFree(echo_h); (void)hipFree(relay); (void)hipStreamDestroy(st);
    return 0;
}
#include <signal.h>
static void on_segv(int) { const char m[] = "the CPU cannot store into fine-grained GPU memory on this system (fault): no such variant here\n"; (void)!write(1, m, sizeof m - 1); _exit(0); }
int main(int argc, char **argv) {
    const int Ps[4] = {1, 16, 64, 144};
    if (argc > 1 && !strcmp(argv[1], "vram")) {
        signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
        for (int p = 0; p < 4; p++) { if (measure(Ps[p], 0, 0, 2, 256, 0, 1) == 1) return 1; if (measure(Ps[p], 0, 0, 2, 1024, 96 * 1024, 1) == 1) return 1; if (measure(Ps[p], 1, 0, 2, 1024, 96 * 1024, 0) == 1) return 1; }
        return 0;
    }
    for (int p = 0; p < 4; p++) for (int mode = 0; mode < 3; mode++) { int rc = measure(Ps[p], mode == 2 ? 1 : 0, mode == 1 ? 1 : 0); if (rc == 1) return 1; }
    // the leader-and-relay form with what a round of the persistent sum-check tail mails, small and tail-sized workgroups
    for (int p = 0; p < 4; p++) { if (measure(Ps[p], 1, 0, 1) == 1) return 1; if (measure(Ps[p], 1, 0, 1, 1024, 96 * 1024) == 1) return 1; if (measure(Ps[p], 1, 0, 2) == 1) return 1; if (measure(Ps[p], 1, 0, 2, 1024, 96 * 1024) == 1) return 1; }
    return 0;
}
