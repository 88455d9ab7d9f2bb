// Latency of ONE Keccak-f[1600] permutation on gfx950, the way a device-side Merlin transcript would have to run it: a single
// dependent chain (every challenge hashes the previous message), so what counts is the time of one permutation on one wave, not
// throughput.  Two shapes:
//   one_lane   the whole state in the registers of one lane (25 x u64), the textbook round
//   lanes25    one lane per state word: theta's column parities, pi's transposition and chi's row neighbours go through LDS (two
//              write -> read exchanges per round; the wave is in lock step, so a workgroup barrier of one wave costs nothing beyond
//              the LDS round trip)
// Both are checked against a host permutation.  Prints nanoseconds per permutation (dependent chain of `iters` permutations).
//   hipcc -O3 --offload-arch=gfx950 tools/keccakbench.hip -o tools/keccakbench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

static const uint64_t RC_H[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
                                  0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                  0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
                                  0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
// rotation offsets r[x + 5 y]
static const int RHO_H[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
__constant__ uint64_t RC[24];
__constant__ int RHO[25];

static inline uint64_t rotl_h(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
static void keccak_host(uint64_t a[25]) {
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) { uint64_t d = c[(x + 4) % 5] ^ rotl_h(c[(x + 1) % 5], 1); for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d; }
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl_h(a[x + 5 * y], RHO_H[x + 5 * y]);
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC_H[rnd];
    }
}

__device__ __forceinline__ uint64_t rotl(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

__global__ __launch_bounds__(64) void k_one_lane(uint64_t *st, int iters) {
    if (threadIdx.x) return;
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = st[i];
    for (int it = 0; it < iters; it++) {
#pragma unroll 1
        for (int rnd = 0; rnd < 24; rnd++) {
            uint64_t c[5], b[25];
#pragma unroll
            for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
            for (int x = 0; x < 5; x++) {
                const uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
#pragma unroll
                for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
            }
            // rho offsets as literals (unrolled): constant-amount rotates are two v_alignbit_b32 each
            constexpr int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll
            for (int x = 0; x < 5; x++)
#pragma unroll
                for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], R[x + 5 * y]);
#pragma unroll
            for (int x = 0; x < 5; x++)
#pragma unroll
                for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
            a[0] ^= RC[rnd];
        }
    }
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = a[i];
}

__global__ __launch_bounds__(64) void k_lanes25(uint64_t *st, int iters) {
    __shared__ uint64_t A[32], B[32];
    const int t = threadIdx.x, x = t % 5, y = (t / 5) % 5;
    const bool on = t < 25;
    uint64_t a = on ? st[t] : 0;
    const int rho = on ? RHO[t] : 0, dst = y + 5 * ((2 * x + 3 * y) % 5), xm = (x + 4) % 5, xp = (x + 1) % 5, n1 = (x + 1) % 5 + 5 * y, n2 = (x + 2) % 5 + 5 * y;
    for (int it = 0; it < iters; it++) {
#pragma unroll 1
        for (int rnd = 0; rnd < 24; rnd++) {
            if (on) A[t] = a;
            __syncthreads();
            const uint64_t cm = A[xm] ^ A[xm + 5] ^ A[xm + 10] ^ A[xm + 15] ^ A[xm + 20];
            const uint64_t cp = A[xp] ^ A[xp + 5] ^ A[xp + 10] ^ A[xp + 15] ^ A[xp + 20];
            a ^= cm ^ rotl(cp, 1);
            if (on) B[dst] = rotl(a, rho);
            __syncthreads();
            const uint64_t b0 = B[t & 31], b1 = B[n1], b2 = B[n2];
            a = b0 ^ (~b1 & b2);
            if (t == 0) a ^= RC[rnd];
        }
    }
    if (on) st[t] = a;
}

int main() {
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(RC), RC_H, sizeof RC_H));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(RHO), RHO_H, sizeof RHO_H));
    uint64_t h0[25], want[25], got[25];
    for (int i = 0; i < 25; i++) h0[i] = 0x9e3779b97f4a7c15ULL * (uint64_t)(i + 1);
    const int iters = 2000;
    memcpy(want, h0, sizeof want);
    const auto c0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++) keccak_host(want);
    const double host_ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - c0).count() / iters;
    printf("host (plain C, one core)           %8.1f ns per permutation\n", host_ns);
    uint64_t *d; CHECK(hipMalloc((void **)&d, sizeof h0));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; variant++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemcpy(d, h0, sizeof h0, hipMemcpyHostToDevice));
            CHECK(hipEventRecord(e0, 0));
            if (variant == 0) hipLaunchKernelGGL(k_one_lane, 1, 64, 0, 0, d, iters); else hipLaunchKernelGGL(k_lanes25, 1, 64, 0, 0, d, iters);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CHECK(hipMemcpy(got, d, sizeof got, hipMemcpyDeviceToHost));
        const bool ok = memcmp(got, want, sizeof want) == 0;
        printf("%-34s %8.1f ns per permutation   %s\n", variant == 0 ? "gfx950 one lane (25 x u64 in VGPRs)" : "gfx950 25 lanes (LDS exchanges)", 1e6 * best / iters, ok ? "matches the host" : "MISMATCH");
        if (!ok) return 2;
    }
    return 0;
}
