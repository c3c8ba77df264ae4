// micro-benchmark: the read-only streaming ceiling of this GPU -- a plain sum over a buffer much larger than the
// 256 MiB Infinity Cache, 16-byte loads, with and without the nontemporal hint, over a range of grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k_sum(const double* __restrict__ p, size_t n2, double* out) {
    double s = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        double2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const double* q = p + 2 * (i + u * stride);
            if (NT) { v[u].x = __builtin_nontemporal_load(q); v[u].y = __builtin_nontemporal_load(q + 1); }
            else v[u] = *reinterpret_cast<const double2*>(q);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) { s += p[2 * i] + p[2 * i + 1]; }
    if (s == 12345.678) out[0] = s;
}
template <bool NT, int UNROLL>
void run(const double* p, size_t n, double* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int bpc : {4, 8, 16, 32}) {
        const int blocks = 256 * bpc;
        hipLaunchKernelGGL((k_sum<NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, p, n / 2, out);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_sum<NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, p, n / 2, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("nt=%d unroll=%d blocks/CU=%2d: %.0f GB/s\n", (int)NT, UNROLL, bpc, 5.0 * n * 8 / ms * 1e-6);
    }
}
int main() {
    const size_t n = (size_t)1 << 28;      // 2 GiB of doubles
    double *p, *out; (void)hipMalloc(&p, n * 8); (void)hipMalloc(&out, 8);
    (void)hipMemset(p, 0, n * 8);
    run<false, 4>(p, n, out); run<true, 4>(p, n, out); run<false, 8>(p, n, out); run<true, 8>(p, n, out);
    return 0;
}
