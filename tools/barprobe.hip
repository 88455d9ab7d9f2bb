// Can a host thread store straight into HBM (fine-grained device memory through the PCIe BAR), and what does a host -> device -> host
// round trip cost when the device polls (a) pinned host memory, (b) that device-memory word?  One persistent wave spins on a sequence
// word and echoes it into pinned host memory; the host measures the time from its store to seeing the echo.
// The CPU store is first tried under a fault handler.
//   hipcc -O2 --offload-arch=gfx950 tools/barprobe.hip -o tools/barprobe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <signal.h>
#include <setjmp.h>
#include <chrono>
#include <algorithm>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %s (%d) at line %d\n", hipGetErrorString(r_), (int)r_, __LINE__); return 1; } } while (0)

__global__ void k_echo(const unsigned long long *door, unsigned long long *echo, unsigned long long rounds, int system_scope) {
    if (threadIdx.x) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned long long want = 1; want <= rounds; want++) {
        for (;;) {
            const unsigned long long s = system_scope ? __hip_atomic_load(door, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s >= want) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) return;      // 5 s: never hang the box
        }
        __hip_atomic_store(echo, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }

static int measure(const char *what, unsigned long long *door_host_view, unsigned long long *door_dev_view, int system_scope) {
    unsigned long long *echo_h = nullptr, *echo_d = nullptr;
    CHECK(hipHostMalloc((void **)&echo_h, 64, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void **)&echo_d, echo_h, 0));
    *echo_h = 0; *(volatile unsigned long long *)door_host_view = 0;
    const unsigned long long rounds = 2000;
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_echo, 1, 64, 0, st, (const unsigned long long *)door_dev_view, echo_d, rounds, system_scope);
    usleep(20000);
    std::vector<double> us; us.reserve(rounds);
    for (unsigned long long r = 1; r <= rounds; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(door_host_view, r, __ATOMIC_RELEASE);
        while (__atomic_load_n(echo_h, __ATOMIC_ACQUIRE) < r) { if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { printf("%-44s echo never came (round %llu)\n", what, r); __atomic_store_n(door_host_view, rounds + 1, __ATOMIC_RELEASE); (void)hipStreamSynchronize(st); return 2; } }
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    CHECK(hipStreamSynchronize(st));
    std::sort(us.begin(), us.end());
    printf("%-44s round trip: median %.2f us, p10 %.2f, p90 %.2f\n", what, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
    (void)hipHostFree(echo_h); (void)hipStreamDestroy(st);
    return 0;
}

int main() {
    // (a) doorbell in pinned host memory
    unsigned long long *dh = nullptr, *dd = nullptr;
    CHECK(hipHostMalloc((void **)&dh, 64, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void **)&dd, dh, 0));
    if (measure("doorbell in pinned host memory (device polls over PCIe)", dh, dd, 1)) return 1;
    // (b) doorbell in fine-grained device memory, written by the CPU through the BAR
    for (int kind = 0; kind < 2; kind++) {
        unsigned long long *dv = nullptr;
        hipError_t e = kind == 0 ? hipExtMallocWithFlags((void **)&dv, 4096, hipDeviceMallocFinegrained) : hipMallocManaged((void **)&dv, 4096, hipMemAttachGlobal);
        const char *name = kind == 0 ? "doorbell in fine-grained device memory (hipExtMallocWithFlags)" : "doorbell in managed memory, preferred location device";
        if (e != hipSuccess) { printf("%-44s allocation refused: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        if (kind == 1) { (void)hipMemAdvise(dv, 4096, hipMemAdviseSetPreferredLocation, 0); (void)hipMemAdvise(dv, 4096, hipMemAdviseSetAccessedBy, hipCpuDeviceId); (void)hipMemPrefetchAsync(dv, 4096, 0, 0); (void)hipDeviceSynchronize(); }
        CHECK(hipMemset(dv, 0, 4096)); CHECK(hipDeviceSynchronize());
        fflush(stdout);
        // the CPU store is tried under a SIGSEGV / SIGBUS handler (a forked child would not inherit the driver's mapping)
        struct sigaction sa, old_segv, old_bus; memset(&sa, 0, sizeof sa); sa.sa_handler = on_fault; sigemptyset(&sa.sa_mask);
        sigaction(SIGSEGV, &sa, &old_segv); sigaction(SIGBUS, &sa, &old_bus);
        bool faulted = false;
        if (sigsetjmp(g_jmp, 1) == 0) { *(volatile unsigned long long *)dv = 0; (void)*(volatile unsigned long long *)dv; } else faulted = true;
        sigaction(SIGSEGV, &old_segv, nullptr); sigaction(SIGBUS, &old_bus, nullptr);
        if (faulted) { printf("%-44s CPU store faults: not host-accessible on this box\n", name); continue; }
        if (measure(name, dv, dv, kind == 1 ? 1 : 0) == 1) return 1;
    }
    return 0;
}
