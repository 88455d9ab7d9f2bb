// floor of a one-shot GPU process: runtime start, one allocation, one launch, one copy back (compare with `spzk verify`'s wall time)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(unsigned *p) { p[threadIdx.x] = threadIdx.x; }
int main() {
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    unsigned *d; if (hipMalloc(&d, 256) != hipSuccess) return 1;
    printf("first hipMalloc %.1f ms\n", ms());
    k<<<1, 64>>>(d); unsigned h[64]; if (hipMemcpy(h, d, 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("first launch + copy %.1f ms\n", ms());
    void *big; if (hipMalloc(&big, (size_t)52 << 30) != hipSuccess) return 1;
    printf("52 GB hipMalloc %.1f ms\n", ms());
    return h[5] == 5 ? 0 : 1;
}
