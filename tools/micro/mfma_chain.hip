// micro-benchmark: v_mfma_f64_16x16x4 issue rate versus number of independent accumulator chains and waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(double* out, int threads) {
    const int iters = 4096 / CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int blocks = 256 * blocks_per_cu;
        hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (threads / 64) * iters * 8 * CH * 2048.0;
        printf("chains %d  waves/SIMD %d : %.1f TFLOP/s\n", CH, threads / 256 * blocks_per_cu, flops / ms * 1e-9);
    }
}
int main() {
    double* out; hipMalloc(&out, 512 * 1024 * 8);
    for (int threads : {256, 512}) { run<1>(out, threads); run<2>(out, threads); run<4>(out, threads); }
    return 0;
}
