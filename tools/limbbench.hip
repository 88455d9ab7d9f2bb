// Round 4: what nine 29-bit limbs buy on gfx950 — whole-chip throughput of the field products and of the mixed point addition in the
// old forms (field.h fr_mul: 8 x u32 saturated, asm carry chains; fp10.h: ten 26/25-bit limbs) and the new ones (fr9.h, fp9.h), every
// result compared with the old form's, plus per-instruction issue costs of the instructions the products are made of.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iotti_amd/csrc tools/limbbench.hip -o tools/limbbench.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "field.h"
#include "point.h"
#include "fp10.h"
#include "fr9.h"
#include "fp9.h"
using namespace otti;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0 fr_mul (two chains per lane, as otti_bench_fr_mul_peak), 1 fr9_mul (two chains), 2 f10_mul, 3 f9_mul, 4 p10_madd, 5 p9_madd
template <int MODE> __global__ __launch_bounds__(256) void k_chain(Fp *io, int iters) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fp a = io[2 * i], b = io[2 * i + 1];
    if (MODE == 0) {
        Fr x, y, z; for (int q = 0; q < 8; q++) { x.v[q] = a.v[q]; y.v[q] = b.v[q]; } x.v[7] &= 0x0fffffffu; y.v[7] &= 0x0fffffffu; z = y;
        for (int k = 0; k < iters; k++) { x = fr_mul(x, y); z = fr_mul(z, y); }
        x = fr_add(x, z); for (int q = 0; q < 8; q++) a.v[q] = x.v[q];
    }
    if (MODE == 1) {
        Fr x, y, z; for (int q = 0; q < 8; q++) { x.v[q] = a.v[q]; y.v[q] = b.v[q]; } x.v[7] &= 0x0fffffffu; y.v[7] &= 0x0fffffffu; z = y;
        const Fr9 y5 = fr9_unpack5(y); Fr9 x9 = fr9_unpack(x), z9 = fr9_unpack(z);          // x <- x * y in the memory format: mont261(x, 32 y)
        for (int k = 0; k < iters; k++) { x9 = fr9_mul(x9, y5); z9 = fr9_mul(z9, y5); }
        x = fr_add(fr9_pack_lt2l(x9), fr9_pack_lt2l(z9)); for (int q = 0; q < 8; q++) a.v[q] = x.v[q];
    }
    if (MODE == 2) { F10 x = f10_unpack(a), y = f10_unpack(b); for (int k = 0; k < iters; k++) x = f10_mul(x, y); a = f10_pack(x); }
    if (MODE == 3) { F9 x = f9_unpack(a), y = f9_unpack(b); for (int k = 0; k < iters; k++) x = f9_mul(x, y); a = f9_pack(x); }
    if (MODE == 4) { P10 p; p.X = f10_unpack(a); p.Y = f10_unpack(b); p.Z = f10_one(); p.T = f10_mul(p.X, p.Y); N10 n; n.yplusx = p.X; n.yminusx = p.Y; n.xy2d = p.T;
                     for (int k = 0; k < iters; k++) p = p10_madd(p, n); a = f10_pack(f10_add(f10_add(p.X, p.Y), f10_add(p.Z, p.T))); }
    if (MODE == 5) { P9 p; p.X = f9_unpack(a); p.Y = f9_unpack(b); p.Z = f9_one(); p.T = f9_mul(p.X, p.Y); N9 n; n.yplusx = p.X; n.yminusx = p.Y; n.xy2d = p.T;
                     for (int k = 0; k < iters; k++) p = p9_madd(p, n); a = f9_pack(f9_add(f9_add(p.X, p.Y), f9_add(p.Z, p.T))); }
    io[2 * i] = a;
}
static bool fp_same(const Fp &a, const Fp &b) { return fp_eq(a, b); }
template <int MODE> static int run(const char *name, int blocks, int threads, int iters, double ops_per_iter, std::vector<Fp> *out) {
    const size_t n = (size_t)blocks * threads;
    std::vector<Fp> h(2 * n);
    for (size_t i = 0; i < 2 * n; i++) for (int q = 0; q < 8; q++) h[i].v[q] = (uint32_t)(0x9e3779b9u * (i * 8 + q + 1)) ^ (uint32_t)((i * 8 + q) * 0x85ebca6bu >> 7);
    Fp *d; CK(hipMalloc((void **)&d, 2 * n * sizeof(Fp))); CK(hipMemcpy(d, h.data(), 2 * n * sizeof(Fp), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<MODE>, blocks, threads, 0, 0, d, iters); CK(hipDeviceSynchronize());
    if (out) { out->resize(2 * n); CK(hipMemcpy(out->data(), d, 2 * n * sizeof(Fp), hipMemcpyDeviceToHost)); }
    CK(hipMemcpy(d, h.data(), 2 * n * sizeof(Fp), hipMemcpyHostToDevice));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chain<MODE>, blocks, threads, 0, 0, d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-34s blocks=%5d thr=%4d iters=%5d : %8.3f ms  -> %8.1f ns per op per chain, %8.2f Gop/s chip\n", name, blocks, threads, iters, best,
           best * 1e6 / (iters * ops_per_iter), n * (double)iters * ops_per_iter / (best * 1e6));
    CK(hipFree(d)); return 0;
}

// ---- per-instruction issue cost: 8 independent instructions per asm block, 64 blocks per loop trip
#define REP8(s) s s s s s s s s
template <int OP> __global__ __launch_bounds__(256) void k_issue(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;
    const uint32_t y = seed | 1;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile(REP8("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                                       "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\t")
                                  : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(a0), "v"(y) : "vcc");
        if (OP == 1) asm volatile(REP8("v_lshl_add_u64 %0, %1, 0, %0\n\tv_lshl_add_u64 %1, %2, 0, %1\n\tv_lshl_add_u64 %2, %3, 0, %2\n\tv_lshl_add_u64 %3, %4, 0, %3\n\t"
                                       "v_lshl_add_u64 %4, %5, 0, %4\n\tv_lshl_add_u64 %5, %6, 0, %5\n\tv_lshl_add_u64 %6, %7, 0, %6\n\tv_lshl_add_u64 %7, %0, 0, %7\n\t")
                                  : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7));
        if (OP == 2) asm volatile(REP8("v_lshrrev_b64 %0, 29, %1\n\tv_lshrrev_b64 %1, 29, %2\n\tv_lshrrev_b64 %2, 29, %3\n\tv_lshrrev_b64 %3, 29, %4\n\t"
                                       "v_lshrrev_b64 %4, 29, %5\n\tv_lshrrev_b64 %5, 29, %6\n\tv_lshrrev_b64 %6, 29, %7\n\tv_lshrrev_b64 %7, 29, %0\n\t")
                                  : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7));
        if (OP == 3) asm volatile(REP8("v_and_b32 %0, %8, %1\n\tv_and_b32 %1, %8, %2\n\tv_and_b32 %2, %8, %3\n\tv_and_b32 %3, %8, %4\n\t"
                                       "v_and_b32 %4, %8, %5\n\tv_and_b32 %5, %8, %6\n\tv_and_b32 %6, %8, %7\n\tv_and_b32 %7, %8, %0\n\t")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        if (OP == 4) asm volatile(REP8("v_add_u32 %0, %8, %1\n\tv_add_u32 %1, %8, %2\n\tv_add_u32 %2, %8, %3\n\tv_add_u32 %3, %8, %4\n\t"
                                       "v_add_u32 %4, %8, %5\n\tv_add_u32 %5, %8, %6\n\tv_add_u32 %6, %8, %7\n\tv_add_u32 %7, %8, %0\n\t")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        if (OP == 5) asm volatile(REP8("v_mul_lo_u32 %0, %8, %1\n\tv_mul_lo_u32 %1, %8, %2\n\tv_mul_lo_u32 %2, %8, %3\n\tv_mul_lo_u32 %3, %8, %4\n\t"
                                       "v_mul_lo_u32 %4, %8, %5\n\tv_mul_lo_u32 %5, %8, %6\n\tv_mul_lo_u32 %6, %8, %7\n\tv_mul_lo_u32 %7, %8, %0\n\t")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        if (OP == 6) asm volatile(REP8("v_alignbit_b32 %0, %1, %2, 29\n\tv_alignbit_b32 %1, %2, %3, 26\n\tv_alignbit_b32 %2, %3, %4, 23\n\tv_alignbit_b32 %3, %4, %5, 20\n\t"
                                       "v_alignbit_b32 %4, %5, %6, 17\n\tv_alignbit_b32 %5, %6, %7, 14\n\tv_alignbit_b32 %6, %7, %0, 11\n\tv_alignbit_b32 %7, %0, %1, 8\n\t")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 7) asm volatile(REP8("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
                                       "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc\n\t")
                                  : "+v"(w0), "+v"(a1), "+v"(w2), "+v"(a3), "+v"(w4), "+v"(a5), "+v"(w6), "+v"(a7) : "v"(a0), "v"(y) : "vcc");
        if (OP == 8) asm volatile(REP8("v_lshl_or_b32 %0, %1, 5, %2\n\tv_lshl_or_b32 %1, %2, 5, %3\n\tv_lshl_or_b32 %2, %3, 5, %4\n\tv_lshl_or_b32 %3, %4, 5, %5\n\t"
                                       "v_lshl_or_b32 %4, %5, 5, %6\n\tv_lshl_or_b32 %5, %6, 5, %7\n\tv_lshl_or_b32 %6, %7, 5, %0\n\tv_lshl_or_b32 %7, %0, 5, %1\n\t")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 9) asm volatile(REP8("v_mad_i64_i32 %0, vcc, %8, %9, %0\n\tv_mad_i64_i32 %1, vcc, %8, %9, %1\n\tv_mad_i64_i32 %2, vcc, %8, %9, %2\n\tv_mad_i64_i32 %3, vcc, %8, %9, %3\n\t"
                                       "v_mad_i64_i32 %4, vcc, %8, %9, %4\n\tv_mad_i64_i32 %5, vcc, %8, %9, %5\n\tv_mad_i64_i32 %6, vcc, %8, %9, %6\n\tv_mad_i64_i32 %7, vcc, %8, %9, %7\n\t")
                                  : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(a0), "v"(y) : "vcc");
        if (OP == 10) asm volatile(REP8("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                                        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t")
                                   : "+v"(w0) : "v"(a0), "v"(y) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7);
}
template <int OP> static int issue(const char *name, int blocks_per_cu, uint32_t *d) {
    const int blocks = 256 * blocks_per_cu, threads = 256, iters = 2000; const double per_trip = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<OP>, blocks, threads, 0, 0, d, iters, 12345u); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_issue<OP>, blocks, threads, 0, 0, d, iters, 12345u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double lane_ops = (double)blocks * threads * iters * per_trip;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal: 1024 SIMDs, waves = blocks * 4
    const double wave_instrs_per_simd = (double)blocks * 4 / 1024.0 * iters * per_trip;
    printf("%-26s %2d waves/SIMD : %8.3f ms  %7.2f T lane-ops/s  %5.2f nominal cycles per wave-instruction\n", name, blocks_per_cu, best, lane_ops / best / 1e9,
           best * 1e-3 * 2.4e9 / wave_instrs_per_simd);
    return 0;
}

int main() {
    uint32_t *d; CK(hipMalloc((void **)&d, (size_t)256 * 8 * 256 * 4));
    for (int occ = 1; occ <= 8; occ *= 2) {
        issue<0>("v_mad_u64_u32", occ, d); issue<10>("v_mad_u64_u32 dependent", occ, d); issue<9>("v_mad_i64_i32", occ, d); issue<7>("v_mad_u64_u32+v_addc", occ, d); issue<1>("v_lshl_add_u64", occ, d);
        issue<2>("v_lshrrev_b64", occ, d); issue<3>("v_and_b32", occ, d); issue<4>("v_add_u32", occ, d); issue<5>("v_mul_lo_u32", occ, d);
        issue<6>("v_alignbit_b32", occ, d); issue<8>("v_lshl_or_b32", occ, d);
    }
    std::vector<Fp> r0, r1, r2, r3, r4, r5;
    run<0>("fr_mul  (8x32, two chains)", 2048, 256, 500, 2, &r0);
    run<1>("fr9_mul (9x29, two chains)", 2048, 256, 500, 2, &r1);
    run<2>("f10_mul (10x26/25)", 2048, 256, 500, 1, &r2);
    run<3>("f9_mul  (9x29)", 2048, 256, 500, 1, &r3);
    run<4>("p10_madd", 1024, 256, 200, 1, &r4);
    run<5>("p9_madd", 1024, 256, 200, 1, &r5);
    run<4>("p10_madd 2048 blocks", 2048, 256, 200, 1, nullptr);
    run<5>("p9_madd 2048 blocks", 2048, 256, 200, 1, nullptr);
    size_t bad_fr = 0, bad_fp = 0, bad_pt = 0;
    for (size_t i = 0; i < r0.size(); i += 2) {
        if (memcmp(&r0[i], &r1[i], 32)) bad_fr++;
        if (!fp_same(r2[i], r3[i])) bad_fp++;
    }
    for (size_t i = 0; i < r4.size(); i += 2) if (!fp_same(r4[i], r5[i])) bad_pt++;
    printf("fr9 vs fr mismatches: %zu of %zu; f9 vs f10: %zu; p9_madd vs p10_madd: %zu of %zu\n", bad_fr, r0.size() / 2, bad_fp, bad_pt, r4.size() / 2);
    return (bad_fr || bad_fp || bad_pt) ? 1 : 0;
}
