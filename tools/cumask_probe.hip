// does a CU-masked stream keep its kernels off the excluded CUs?  A long-running kernel fills the chip from the masked stream; a tiny kernel
// on another stream is timed meanwhile.  HW_REG_XCC_ID / HW_ID per workgroup are recorded to see where the masked kernel ran.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#include <set>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)
__global__ void k_busy(unsigned *where, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); where[blockIdx.x] = (xcc << 16) | (hw & 0xffff); }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
}
__global__ void k_tiny(unsigned *x) { if (threadIdx.x == 0) x[0] += 1; }
int main() {
    hipStream_t masked = nullptr, plain = nullptr, other = nullptr;
    uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = i == 0 ? 0u : 0xffffffffu;
    hipError_t e = hipExtStreamCreateWithCUMask(&masked, 8, mask);
    printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
    unsigned *where, *x; CK(hipMalloc(&where, 4096 * 4)); CK(hipMalloc(&x, 4)); CK(hipMemset(x, 0, 4));
    for (int variant = 0; variant < 2; variant++) {
        hipStream_t busy = variant == 0 ? plain : masked;
        if (!busy) continue;
        // 2048 workgroups of 1024 threads with 64 KB of LDS each would be the harshest; here: 4096 x 256 threads, each busy for 2 ms
        hipLaunchKernelGGL(k_busy, 4096, 256, 0, busy, where, 200000ull);

        auto t0 = std::chrono::steady_clock::now();
        double worst = 0, sum = 0; int n = 0;
        while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(6)) {
            auto a = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_tiny, 1, 64, 0, other, x); CK(hipStreamSynchronize(other));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
            worst = us > worst ? us : worst; sum += us; n++;
        }
        CK(hipDeviceSynchronize());
        unsigned h[4096]; CK(hipMemcpy(h, where, sizeof h, hipMemcpyDeviceToHost));
        std::set<unsigned> cus, xccs; for (unsigned v : h) { cus.insert(v); xccs.insert(v >> 16); }
        printf("%s stream busy: tiny kernel on another stream: %d launches, mean %.1f us, worst %.1f us; busy kernel touched %zu distinct (xcc, hw_id) and %zu XCCs\n",
               variant == 0 ? "plain " : "masked", n, sum / n, worst, cus.size(), xccs.size());
    }
    return 0;
}
