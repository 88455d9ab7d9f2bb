// floor of a one-shot GPU process: runtime start, one allocation, one launch, one copy back, and — as spzk does — _exit without the
// runtime's teardown (compare with `spzk verify`'s wall time).  `hipfloor.bin teardown` returns from main instead.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>
__global__ void k(unsigned *p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv) {
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    unsigned *d; if (hipMalloc(&d, 256) != hipSuccess) return 1;
    printf("first hipMalloc %.1f ms\n", ms());
    k<<<1, 64>>>(d); unsigned h[64]; if (hipMemcpy(h, d, 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("first launch + copy %.1f ms\n", ms());
    if (argc > 1 && !strcmp(argv[1], "teardown")) return h[5] == 5 ? 0 : 1;
    fflush(stdout); _exit(h[5] == 5 ? 0 : 1);
}
