// Microbenchmark of the gfx950 field multiplies: single-wave dependent-chain latency and full-chip throughput.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I otti_amd/csrc tools/mulbench.hip -o /tmp/mulbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "field.h"
#include "point.h"
#include "fp10.h"
using namespace otti;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE> __global__ void k_chain(Fp *io, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fp a = io[2 * i], b = io[2 * i + 1];
    if (MODE == 0) for (int k = 0; k < iters; k++) a = fp_mul(a, b);
    if (MODE == 1) { Fr x, y; for (int q = 0; q < 8; q++) { x.v[q] = a.v[q] & 0x0fffffffu; y.v[q] = b.v[q] & 0x0fffffffu; } for (int k = 0; k < iters; k++) x = fr_mul(x, y); for (int q = 0; q < 8; q++) a.v[q] = x.v[q]; }
    if (MODE == 2) for (int k = 0; k < iters; k++) a = fp_add(fp_sub(a, b), a);
    if (MODE == 3) { Pt p; p.X = a; p.Y = b; p.Z = fp_one(); p.T = fp_mul(a, b); Pt q = p; for (int k = 0; k < iters; k++) p = pt_add(p, q); a = p.X; }
    if (MODE == 4) { Pt p; p.X = a; p.Y = b; p.Z = fp_one(); p.T = fp_mul(a, b); Niels n; n.yplusx = a; n.yminusx = b; n.xy2d = p.T; for (int k = 0; k < iters; k++) p = pt_madd(p, n); a = p.X; }
    if (MODE == 5) { F10 x = f10_unpack(a), y = f10_unpack(b); for (int k = 0; k < iters; k++) x = f10_mul(x, y); a = f10_pack(x); }
    if (MODE == 6) { F10 x = f10_unpack(a); for (int k = 0; k < iters; k++) x = f10_sqr(x); a = f10_pack(x); }
    if (MODE == 7) { P10 p; p.X = f10_unpack(a); p.Y = f10_unpack(b); p.Z = f10_one(); p.T = f10_mul(p.X, p.Y); N10 n; n.yplusx = p.X; n.yminusx = p.Y; n.xy2d = p.T; for (int k = 0; k < iters; k++) p = p10_madd(p, n); a = f10_pack(p.X); }
    if (MODE == 8) { P10 p; p.X = f10_unpack(a); p.Y = f10_unpack(b); p.Z = f10_one(); p.T = f10_mul(p.X, p.Y); P10 q = p; F10 d2 = f10_const(fp_2D()); for (int k = 0; k < iters; k++) p = p10_add(p, q, d2); a = f10_pack(p.X); }
    // quad-parallel forms (fp10.h): four lanes per point; one "op" = one addition of the whole quad
    if (MODE == 9) { const int q = threadIdx.x & 3; F10 acc = f10_unpack(a), v = f10_unpack(b); for (int k = 0; k < iters; k++) acc = q10_add_cached(acc, v, q); a = f10_pack(acc); }
    if (MODE == 10) { const int q = threadIdx.x & 3; F10 acc = f10_unpack(a), o = f10_unpack(b); const F10 d2 = f10_const(fp_2D());
                      for (int k = 0; k < iters; k++) acc = q10_add_cached(acc, q10_cached(q10_u(o, q), q, d2), q), o = acc; a = f10_pack(acc); }
    if (MODE == 11) { const int q = threadIdx.x & 3; F10 acc = f10_unpack(a); for (int k = 0; k < iters; k++) acc = q10_u(acc, q); a = f10_pack(acc); }
    io[2 * i] = a;
}
template <int MODE> static int run(const char *name, int blocks, int threads, int iters, double ops_per_iter) {
    size_t n = (size_t)blocks * threads;
    std::vector<Fp> h(2 * n);
    for (size_t i = 0; i < 2 * n; i++) for (int q = 0; q < 8; q++) h[i].v[q] = (uint32_t)(0x9e3779b9u * (i * 8 + q + 1));
    Fp *d; CK(hipMalloc((void **)&d, 2 * n * sizeof(Fp))); CK(hipMemcpy(d, h.data(), 2 * n * sizeof(Fp), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<MODE>, blocks, threads, 0, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chain<MODE>, blocks, threads, 0, 0, d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s blocks=%5d thr=%4d iters=%5d : %8.3f ms  -> %8.1f ns per op per thread-chain, %8.2f Gop/s chip\n", name, blocks, threads, iters, ms,
           ms * 1e6 / (iters * ops_per_iter), n * (double)iters * ops_per_iter / (ms * 1e6));
    CK(hipFree(d)); return 0;
}
int main() {
    run<0>("fp_mul latency (1 wave)", 1, 64, 2000, 1);
    run<1>("fr_mul latency (1 wave)", 1, 64, 2000, 1);
    run<2>("fp_add+sub latency", 1, 64, 2000, 2);
    run<3>("pt_add latency (1 wave)", 1, 64, 500, 1);
    run<4>("pt_madd latency (1 wave)", 1, 64, 500, 1);
    run<5>("f10_mul latency (1 wave)", 1, 64, 2000, 1);
    run<6>("f10_sqr latency (1 wave)", 1, 64, 2000, 1);
    run<7>("p10_madd latency (1 wave)", 1, 64, 500, 1);
    run<8>("p10_add latency (1 wave)", 1, 64, 500, 1);
    run<9>("q10 madd latency (1 wave)", 1, 64, 500, 1);
    run<10>("q10 cached+add latency", 1, 64, 500, 1);
    run<11>("q10_u latency (1 wave)", 1, 64, 2000, 1);
    run<5>("f10_mul throughput", 256 * 8, 256, 500, 1);
    run<6>("f10_sqr throughput", 256 * 8, 256, 500, 1);
    run<7>("p10_madd throughput", 256 * 4, 256, 200, 1);
    run<8>("p10_add throughput", 256 * 4, 256, 200, 1);
    run<0>("fp_mul throughput", 256 * 8, 256, 500, 1);
    run<1>("fr_mul throughput", 256 * 8, 256, 500, 1);
    run<4>("pt_madd throughput", 256 * 4, 256, 200, 1);
    run<3>("pt_add throughput", 256 * 4, 256, 200, 1);
    return 0;
}
