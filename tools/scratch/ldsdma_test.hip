// checks the lane -> LDS address mapping of global_load_lds_dwordx4 on gfx950: lane l of an instruction lands at base + 16 l
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4 *src, const unsigned *idx, uint4 *out) {
    __shared__ uint4 buf[256 * 6];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 *p = src + (size_t)idx[threadIdx.x] * 6;
    for (int k = 0; k < 6; k++)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(p + k), (void __attribute__((address_space(3))) *)(buf + wave * 384 + k * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = 0; k < 6; k++) out[threadIdx.x * 6 + k] = buf[wave * 384 + k * 64 + lane];
}
int main() {
    const int N = 4096;
    std::vector<uint4> h(N * 6); for (int i = 0; i < N * 6; i++) h[i] = {(unsigned)i, (unsigned)i * 3u, 7u, (unsigned)~i};
    std::vector<unsigned> idx(256); for (int i = 0; i < 256; i++) idx[i] = (i * 977u + 13u) % N;
    uint4 *d, *o; unsigned *di;
    hipMalloc(&d, h.size() * 16); hipMalloc(&o, 256 * 6 * 16); hipMalloc(&di, 1024);
    hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice); hipMemcpy(di, idx.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 256>>>(d, di, o);
    std::vector<uint4> r(256 * 6); hipMemcpy(r.data(), o, r.size() * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) for (int k = 0; k < 6; k++) { uint4 e = h[idx[t] * 6 + k], g = r[t * 6 + k]; if (e.x != g.x || e.y != g.y || e.z != g.z || e.w != g.w) bad++; }
    printf("mismatches: %d of %d\n", bad, 256 * 6);
    return bad != 0;
}
