// micro-benchmark: do fp64 MFMA (one wave) and fp64 VALU (another wave of the same SIMD) overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// mode bit 0: even waves run MFMA chains; bit 1: odd waves run VALU FMA chains; 512 threads = 2 waves per SIMD
__global__ __launch_bounds__(512) void k(double* out, int iters, int mode, double a0) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if ((wave & 4) == 0) {          // waves 0-3: one per SIMD
        if (mode & 1) {
            d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            double a = a0 + threadIdx.x, b = a0;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
                }
            }
            s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
        }
    } else {                        // waves 4-7: the second wave of each SIMD
        if (mode & 2) {
            double x0 = a0 + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
            const double m = 1.0000001, c = 1e-9;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int r = 0; r < 32; ++r) {     // 256 FMAs = 1024 cycles, like 16 MFMAs
                    x0 = fma(x0, m, c); x1 = fma(x1, m, c); x2 = fma(x2, m, c); x3 = fma(x3, m, c);
                    x4 = fma(x4, m, c); x5 = fma(x5, m, c); x6 = fma(x6, m, c); x7 = fma(x7, m, c);
                }
            }
            s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* out; (void)hipMalloc(&out, 256 * 512 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode : {1, 2, 3, 1, 2, 3}) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode, 1.0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.3f ms   (ideal alone: %.3f ms at 2.4 GHz)\n", mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms,
               iters * 1024.0 / 2.4e6);
    }
    return 0;
}
