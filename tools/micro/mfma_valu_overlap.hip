// micro-benchmark: do fp64 MFMA (one wave) and fp64 VALU (another wave of the same SIMD) overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// mode bit 0: even waves run MFMA chains; bit 1: odd waves run VALU FMA chains; 512 threads = 2 waves per SIMD
// `rnd`: 512 random doubles in [0.5, 1.5): with them the operands have full-entropy mantissas (the chip's clock under
// load depends on how many bits toggle, MI355X_MICROARCH.md "DVFS give-back"); NULL: small integers as before
__global__ __launch_bounds__(512) void k(double* out, int iters, int mode, double a0, const double* rnd) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (rnd) a0 = rnd[threadIdx.x];
    if ((wave & 4) == 0) {          // waves 0-3: one per SIMD
        if (mode & 1) {
            d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            double a = rnd ? a0 : a0 + threadIdx.x, b = rnd ? rnd[(threadIdx.x * 7 + 3) & 511] : a0;
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
    const int iters = 40000;        // ~17 ms per launch: long enough for the clock to settle
    double h[512];
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < 512; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = 0.5 + (double)(x >> 11) / 9007199254740992.0; }
    double* rnd; (void)hipMalloc(&rnd, sizeof h); (void)hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass)
        for (int mode : {1, 2, 3, 1, 2, 3}) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode, 1.0, pass ? rnd : (const double*)nullptr);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double mfma_tf = (mode & 1) ? 256.0 * 4 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12 : 0.0;
            printf("%s operands, mode %d (%s): %.3f ms   (ideal alone: %.3f ms at 2.4 GHz)%s", pass ? "random-mantissa" : "small-integer", mode,
                   mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms, iters * 1024.0 / 2.4e6, mode & 1 ? "" : "\n");
            if (mode & 1) printf("   fp64 MFMA %.1f TFLOP/s\n", mfma_tf);
        }
    return 0;
}
