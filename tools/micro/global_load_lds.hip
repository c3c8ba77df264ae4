// micro-test: global_load_lds_dwordx4 on gfx950 -- where do a wave's 64 x 16 bytes land in LDS, and do per-lane global addresses
// work (a swizzled gather into a linear LDS tile)?  hipcc -O3 --offload-arch=gfx950 -o gll tools/micro/global_load_lds.hip && ./gll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ src, const int* __restrict__ perm, double* __restrict__ out) {
    __shared__ double buf[4][128];                 // per wave: 64 lanes x 2 doubles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane l fetches the two doubles at src[2 * perm[l]], its 16 bytes should land at buf[wave][2 * l]
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * perm[lane] + 1000 * wave),
                                     (__attribute__((address_space(3))) void*)&buf[wave][0], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);                 // vmcnt(0) among the others
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    out[threadIdx.x * 2] = buf[wave][2 * lane];
    out[threadIdx.x * 2 + 1] = buf[wave][2 * lane + 1];
}
int main() {
    std::vector<double> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    std::vector<int> p(64);
    for (int i = 0; i < 64; ++i) p[i] = (i * 37 + 5) % 64;
    double *d, *o; int* dp;
    hipMalloc(&d, 8192 * 8); hipMalloc(&o, 512 * 8); hipMalloc(&dp, 64 * 4);
    hipMemcpy(d, h.data(), 8192 * 8, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, dp, o);
    std::vector<double> r(512);
    hipMemcpy(r.data(), o, 512 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63, wave = t >> 6;
        const double want0 = 2 * p[lane] + 1000 * wave, want1 = want0 + 1;
        if (r[2 * t] != want0 || r[2 * t + 1] != want1) { if (bad < 5) printf("thread %d: got %g %g want %g %g\n", t, r[2 * t], r[2 * t + 1], want0, want1); ++bad; }
    }
    printf("%s (%d mismatches of 256 threads)\n", bad ? "LAYOUT DIFFERS" : "lane l -> LDS base + 16 l: ok", bad);
    return bad != 0;
}
