// micro-benchmark: throughput and accuracy of candidate device logarithms (fdlibm-style with IEEE division,
// with rcp + Newton, and table-driven with the table in LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../blueice_amd/csrc/bi_log_table.h"

namespace {
__device__ __forceinline__ double log_a(double x) {   // current
    if (!(x >= 2.2250738585072014e-308 && x < __builtin_inf())) return log(x);
    double m = __builtin_amdgcn_frexp_mant(x);
    int k = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752440) { m += m; k -= 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
__device__ __forceinline__ double log_b(double x) {   // rcp + Newton instead of the IEEE division
    if (!(x >= 2.2250738585072014e-308 && x < __builtin_inf())) return log(x);
    double m = __builtin_amdgcn_frexp_mant(x);
    int k = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752440) { m += m; k -= 1; }
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = fma(fma(-d, s, f), r, s);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
__device__ __forceinline__ double log_c(double x, const double4* __restrict__ tab) {   // table-driven (the library's log_core)
    if (!__builtin_amdgcn_class(x, 0x100)) return log(x);      // anything but a positive normal number
    const unsigned long long ix = __double_as_longlong(x);
    const int hi = (int)(ix >> 32);
    const int t = hi - 0x3FE60000;
    const int i = (t >> 13) & 127;
    const int k = t >> 20;
    const double z = __longlong_as_double(((unsigned long long)(unsigned)(hi - (t & 0xFFF00000)) << 32) | (ix & 0xFFFFFFFFull));
    const double4 e = tab[i];
    const double kd = (double)k;
    const double r = fma(z, e.x, -1.0);
    const double w = fma(kd, kLn2Hi, e.y);         // exact: both on grids (bi_log_table.h)
    const double tail = fma(kd, kLn2Lo, e.z);
    double p = fma(r, -1.0 / 8.0, 1.0 / 7.0);
    p = fma(r, p, -1.0 / 6.0);
    p = fma(r, p, 1.0 / 5.0);
    p = fma(r, p, -1.0 / 4.0);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    const double q = fma(r * r, p, tail);
    const double h = w + r;
    const double err = (w - h) + r;
    return h + (err + q);
}

template <int V>
__global__ __launch_bounds__(256) void k_speed(double* out, int iters, double x0, double step) {
    __shared__ double4 tab[128];
    if (V == 2) { if (threadIdx.x < 128) tab[threadIdx.x] = kLogTable[threadIdx.x]; __syncthreads(); }
    double x = x0 + (blockIdx.x * 256 + threadIdx.x) * step, s = 0.0;
    for (int i = 0; i < iters; ++i) {
        s += V == 0 ? log_a(x) : (V == 1 ? log_b(x) : log_c(x, tab));
        x = fma(x, 1.0009765625, step);
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V>
__global__ __launch_bounds__(256) void k_eval(const double* x, double* out, int n) {
    __shared__ double4 tab[128];
    if (V == 2) { if (threadIdx.x < 128) tab[threadIdx.x] = kLogTable[threadIdx.x]; __syncthreads(); }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = V == 0 ? log_a(x[i]) : (V == 1 ? log_b(x[i]) : log_c(x[i], tab));
}
}  // namespace

template <int V>
void speed(double* out) {
    const int blocks = 256 * 8, iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_speed<V>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.3, 1e-7);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_speed<V>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.3, 1e-7);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d: %.1f Glog/s\n", V, (double)blocks * 256 * iters / ms * 1e-6);
}

template <int V>
void accuracy(const std::vector<double>& x, double* dx, double* dout) {
    const int n = (int)x.size();
    std::vector<double> y(n);
    hipLaunchKernelGGL(k_eval<V>, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, n);
    (void)hipMemcpy(y.data(), dout, n * 8, hipMemcpyDeviceToHost);
    double worst = 0; double at = 0;
    for (int i = 0; i < n; ++i) {
        const long double ref = logl((long double)x[i]);
        const double rd = (double)ref;
        const double ulp = rd == 0 ? 4.9e-324 : fabs(nextafter(rd, INFINITY) - rd);
        const double err = (double)(fabsl((long double)y[i] - ref) / ulp);
        if (err > worst) { worst = err; at = x[i]; }
    }
    printf("variant %d: worst error %.3f ulp at x = %.17g\n", V, worst, at);
}

int main() {
    double* out; (void)hipMalloc(&out, 256 * 8 * 256 * 8);
    speed<0>(out); speed<1>(out); speed<2>(out);
    std::vector<double> x;
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < 4000000; ++i) x.push_back(exp((rnd() - 0.5) * 60.0));      // e^-30 .. e^30
    for (int i = 0; i < 4000000; ++i) x.push_back(0.5 + rnd() * 1.5);              // around 1
    for (int i = 0; i < 1000000; ++i) x.push_back(1.0 + (rnd() - 0.5) * 0.03);     // close to 1
    for (int i = 0; i < 1000000; ++i) x.push_back(1.0 + (rnd() - 0.5) * 1e-6);
    for (int i = -1000; i < 1000; ++i) { x.push_back(ldexp(1.0, i)); x.push_back(nextafter(ldexp(1.0, i), 0.0)); x.push_back(ldexp(1.375, i)); x.push_back(nextafter(ldexp(1.375, i), 0.0)); }
    double *dx, *dout; (void)hipMalloc(&dx, x.size() * 8); (void)hipMalloc(&dout, x.size() * 8);
    (void)hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    accuracy<0>(x, dx, dout); accuracy<1>(x, dx, dout); accuracy<2>(x, dx, dout);
    return 0;
}
