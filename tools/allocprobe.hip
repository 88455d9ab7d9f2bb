// What does hipMalloc of a window table cost?  One allocation of S GB against the same bytes in 16 GB pieces, fresh process each (argv: GB, pieces).
//   hipcc -O2 --offload-arch=gfx950 tools/allocprobe.hip -o tools/allocprobe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 96.0; const int pieces = argc > 2 ? atoi(argv[2]) : 1;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    void *warm; if (hipMalloc(&warm, 256) != hipSuccess) return 1;
    const size_t each = (size_t)(gb * 1e9 / pieces);
    std::vector<void *> p(pieces);
    const double t0 = now();
    for (int i = 0; i < pieces; i++) if (hipMalloc(&p[i], each) != hipSuccess) { printf("hipMalloc failed at piece %d\n", i); return 1; }
    const double t1 = now();
    hipMemset(p[0], 0, 4096); hipDeviceSynchronize();
    const double t2 = now();
    for (auto q : p) hipFree(q);
    const double t3 = now();
    printf("%.0f GB in %d piece(s): hipMalloc %.1f ms, first touch %.1f ms, hipFree %.1f ms\n", gb, pieces, t1 - t0, t2 - t1, t3 - t2);
    return 0;
}
