// libblueice_hip: MI355X (gfx950 / CDNA4) binned-likelihood hot path behind a C ABI.
//
// What runs here, per evaluation (reference: JelleAalbers/blueice v1.2.1):
//   a3  GridInterpolator multilinear morph over the anchor tensor   blueice/pdf_morphers.py:57-70
//   a4  rate scaling + early exits                                    blueice/likelihood.py:345-415
//   a6  Beeston-Barlow single-source adjustment                       blueice/likelihood.py:618-660,693-712
//   a5  sum_bins poisson.logpmf(n | sum_s r_s p_s)                    blueice/likelihood.py:662-675
//
// Design: the anchor tensor lives in HBM as rows [anchor][source][Bp] (bin
// fastest, rows padded to a multiple of the 512-bin block tile, so every lane issues aligned
// 16-byte loads with no bounds checks).  One kernel streams the 2^d * S corner rows of a grid
// cell once, and for up to G parameter points that fall in that cell keeps
//     mu[g][bin] = sum_{corner,source} (w_corner[g] * r_source[g]) * row[corner,source][bin]
// in registers (the per-point coefficients are wave-uniform and arrive through scalar loads),
// applies the Poisson term, and reduces wave -> block -> partial.  A small second kernel sums
// the per-block partials in a fixed order (bitwise reproducible; no float atomics) and
// subtracts the per-dataset sum of lgamma(n+1), which depends on the data only and is computed
// once at upload.  The path is HBM-bandwidth bound: 8*(2^d*S + 1) bytes per bin per cell pass.
//
// No CPU fallback exists: without a HIP device bi_create fails.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/blueice_hip.h"

#define BI_VERSION "blueice_hip 0.1 (gfx950)"

namespace {

constexpr int kThreads = 256;           // 4 wave64 per block
constexpr int kBinsPerThread = 2;       // one 16-byte load per stream per lane
constexpr int kTile = kThreads * kBinsPerThread;  // 512 bins = 4 KiB per stream per block tile
constexpr int kMaxDim = 8;              // shape parameters
constexpr int kMaxG = 16;               // points per cell pass

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// ------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ unsigned wave_or(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_down(v, off, 64);
    return v;
}

// Poisson log-pmf without the data-only lgamma(n+1) term, scipy semantics
// (scipy/stats/_distn_infrastructure.py logpmf + _discrete_distns.py poisson._logpmf):
//   mu not >= 0 (negative or nan) or n nan -> nan
//   n negative or non-integer             -> -inf
//   else xlogy(n, mu) - mu                  (xlogy(0, mu) = 0, also for mu = 0)
__device__ __forceinline__ double poisson_term(double n, double mu) {
    double t;
    if (n > 0.0) {
        t = n * log(mu) - mu;  // mu = 0 -> -inf; mu < 0 -> nan
    } else {
        t = -mu;
    }
    if (!(mu >= 0.0) || n != n) t = __builtin_nan("");
    else if (n < 0.0 || n != floor(n)) t = -__builtin_inf();
    return t;
}

// Beeston-Barlow roots, evaluated in the reference's own operation order without FMA
// contraction (blueice/likelihood.py:693-712) so that the sign tests behind its two asserts
// see the same rounding.
__device__ __forceinline__ void bb_roots(double a, double p, double U, double d, double& r1, double& r2) {
#pragma clang fp contract(off)
    double U2 = U * U, p2 = p * p, a2 = a * a, d2 = d * d;
    double disc = U2 * p2 + 2 * U2 * p + U2 + 2 * U * a * p2 + 2 * U * a * p - 2 * U * d * p2 - 2 * U * d * p +
                  a2 * p2 + 2 * a * d * p2 + d2 * p2;
    double lead = -U * p - U + a * p + d * p;
    double den = 2 * p * (p + 1);
    double sq = sqrt(disc);
    r1 = (lead - sq) / den;
    r2 = (lead + sq) / den;
}

struct LaunchArgs {
    const double* ps;       // [rows][Bp]
    const double* nm;       // [A][Bp] (BB) or null
    const double* counts;   // [T][Bp]
    const int64_t* rowoff;  // [items][NS]  element offsets of the stream rows
    const double* coef;     // [items][NS][G]
    const double* aux;      // [items][G][2]  (p_cal, N) for BB
    const int64_t* item_cnt; // [items] element offset of the item's counts row
    const int32_t* item_tiles; // [items] 512-bin tiles of the item's rows (NULL: n_tiles)
    double* partial;        // [items][nbx][G]
    unsigned* pflags;       // [items][nbx][G]
    int64_t B, Bp;
    double outlier;         // MODE 2: likelihood given to events with a non-positive density (0 = none)
    int n0, n1, n2;         // streams into U (or mu), into P_i, into a
    int n_tiles;
};

// The morph + reduce kernel.  blockIdx.y = item (a cell pass with up to G points),
// blockIdx.x strides over 512-bin tiles.
template <bool NT>
__device__ __forceinline__ double2 stream_load(const double* p) {
    if constexpr (NT) {
        // streamed-once data: nontemporal hint (global_load_dwordx4 ... nt) keeps it from displacing L2 / MALL lines
        double2 v;
        v.x = __builtin_nontemporal_load(p);
        v.y = __builtin_nontemporal_load(p + 1);
        return v;
    } else {
        return *reinterpret_cast<const double2*>(p);
    }
}

// MODE 2: as MODE 0 for the extended unbinned likelihood (rows hold pdf values at the events).
// MODE 0: G parameter points of one cell.  MODE 1 (gradient): ONE point; column 0 of the coefficient matrix
// gives mu, columns 1.. give d mu / d theta_j (theta = shape parameters, then rate scales), and the per-bin
// chain rule d ll / d theta_j = (n / mu - 1) * d mu / d theta_j is reduced alongside the likelihood.
template <int G, bool BB, bool NT, int MODE = 0>
__global__ __launch_bounds__(kThreads) void k_morph_reduce(LaunchArgs a) {
    const int item = blockIdx.y;
    const int NS = a.n0 + a.n1 + a.n2;
    const int64_t* __restrict__ rowoff = a.rowoff + (int64_t)item * NS;
    const double* __restrict__ coef = a.coef + (int64_t)item * NS * G;
    const double* __restrict__ cnt = a.counts + a.item_cnt[item];
    const int n_tiles = a.item_tiles ? a.item_tiles[item] : a.n_tiles;

    double sum[G];
    unsigned flg[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { sum[g] = 0.0; flg[g] = 0u; }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double acc[G][2];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g][0] = 0.0; acc[g][1] = 0.0; }

#pragma unroll 8
        for (int k = 0; k < a.n0; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double c = coef[k * G + g];
                acc[g][0] = fma(c, v.x, acc[g][0]);
                acc[g][1] = fma(c, v.y, acc[g][1]);
            }
        }
        double2 nv;
        if constexpr (MODE == 2) { nv.x = nv.y = 0.0; } else { nv = *reinterpret_cast<const double2*>(cnt + bin0); }

        if constexpr (MODE == 2) {
            // extended unbinned likelihood (blueice/likelihood.py:678-690): the "bins" are the events,
            // the term is log(sum_s mu_s p_s(x_e)) with the outlier clamp; -sum_s mu_s is added by the host
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (bin0 + j < a.B) {
                        double lam = acc[g][j];
                        if (a.outlier != 0.0 && !(lam > 0.0)) lam = a.outlier;
                        sum[g] += log(lam);
                    }
                }
            }
        } else if constexpr (MODE == 1) {
            sum[0] += poisson_term(nv.x, acc[0][0]) + poisson_term(nv.y, acc[0][1]);
            const double f0 = (nv.x != 0.0 ? nv.x / acc[0][0] : 0.0) - 1.0;
            const double f1 = (nv.y != 0.0 ? nv.y / acc[0][1] : 0.0) - 1.0;
#pragma unroll
            for (int g = 1; g < G; ++g) sum[g] += f0 * acc[g][0] + f1 * acc[g][1];
        } else if constexpr (!BB) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                sum[g] += poisson_term(nv.x, acc[g][0]) + poisson_term(nv.y, acc[g][1]);
            }
        } else {
            double pi[G][2], ai[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) { pi[g][0] = pi[g][1] = ai[g][0] = ai[g][1] = 0.0; }
#pragma unroll 4
            for (int k = 0; k < a.n1; ++k) {
                const double2 v = stream_load<NT>(a.ps + rowoff[a.n0 + k] + bin0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double c = coef[(a.n0 + k) * G + g];
                    pi[g][0] = fma(c, v.x, pi[g][0]);
                    pi[g][1] = fma(c, v.y, pi[g][1]);
                }
            }
#pragma unroll 4
            for (int k = 0; k < a.n2; ++k) {
                const double2 v = stream_load<NT>(a.nm + rowoff[a.n0 + a.n1 + k] + bin0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double c = coef[(a.n0 + a.n1 + k) * G + g];
                    ai[g][0] = fma(c, v.x, ai[g][0]);
                    ai[g][1] = fma(c, v.y, ai[g][1]);
                }
            }
            const double* __restrict__ aux = a.aux + (int64_t)item * G * 2;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double p_cal = aux[g * 2 + 0];
                const double Ntot = aux[g * 2 + 1];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (bin0 + j < a.B) {
                        const double n = j ? nv.y : nv.x;
                        const double U = acc[g][j];
                        const double ab = ai[g][j];
                        // likelihood.py:645-646
                        const double w = pi[g][j] / ab * Ntot;
                        double r1, r2;
                        bb_roots(ab, w * p_cal, U, n, r1, r2);
                        if (!(r1 <= 0.0)) flg[g] |= BI_ST_BB_ROOT1;
                        const double A = (U == 0.0) ? (n + ab) / (1.0 + p_cal) : r2;
                        if (!(0.0 <= A)) flg[g] |= BI_ST_BB_NEG;
                        const double mu = U + (A * w) * p_cal;
                        sum[g] += poisson_term(n, mu);
                    }
                }
            }
        }
    }

    __shared__ double s_sum[kThreads / 64][G];
    __shared__ unsigned s_flg[kThreads / 64][G];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double s = wave_sum(sum[g]);
        const unsigned f = BB ? wave_or(flg[g]) : 0u;
        if (lane == 0) { s_sum[wave][g] = s; s_flg[wave][g] = f; }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        double s = s_sum[0][g];
        unsigned f = s_flg[0][g];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += s_sum[w][g]; f |= s_flg[w][g]; }
        const int64_t o = ((int64_t)item * gridDim.x + blockIdx.x) * G + g;
        a.partial[o] = s;
        a.pflags[o] = f;
    }
}

// Sum the per-block partials of every (item, g) in a fixed order, subtract the dataset's
// sum lgamma(n+1), scatter to the caller's point order.  `lanes` (64 or 256) threads per slot.
__global__ __launch_bounds__(kThreads) void k_finish(const double* __restrict__ partial,
                                                     const unsigned* __restrict__ pflags, int nbx, int G, int lanes,
                                                     int64_t n_slots, const int64_t* __restrict__ perm,
                                                     const double* __restrict__ slot_lg, double* __restrict__ out,
                                                     int32_t* __restrict__ status) {
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    const int per_block = kThreads / lanes;
    const int64_t slot = (int64_t)blockIdx.x * per_block + threadIdx.x / lanes;
    const int l = threadIdx.x % lanes;
    const bool live = slot < n_slots;
    const int64_t item = live ? slot / G : 0;
    const int g = live ? (int)(slot % G) : 0;
    const int64_t p = live ? perm[slot] : -1;
    const double lg = live ? slot_lg[slot] : 0.0;
    double s = 0.0;
    unsigned f = 0u;
    if (p >= 0) {
#pragma unroll 8
        for (int b = l; b < nbx; b += lanes) {
            const int64_t o = (item * nbx + b) * G + g;
            s += partial[o];
            f |= pflags[o];
        }
    }
    s = wave_sum(s);
    f = wave_or(f);
    if (lanes == 64) {
        if ((threadIdx.x & 63) == 0 && p >= 0) {
            out[p] = s - lg;
            if (status) status[p] |= (int32_t)f;
        }
        return;
    }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; shf[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0 && p >= 0) {
        double t = sh[0];
        unsigned ff = shf[0];
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[w]; ff |= shf[w]; }
        out[p] = t - lg;
        if (status) status[p] |= (int32_t)ff;
    }
}

__global__ void k_fill_const(double* __restrict__ out, const int64_t* __restrict__ idx, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[idx[i]] = v;
}

// sum_b lgamma(n_b + 1) over the valid counts of one dataset chunk -> partial[t][blk]
__global__ __launch_bounds__(kThreads) void k_counts_lgamma(const double* __restrict__ counts, int64_t B, int64_t Bp,
                                                            double* __restrict__ partial, int nblk) {
    const int t = blockIdx.y;
    const double* __restrict__ c = counts + (int64_t)t * Bp;
    double s = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < B; b += (int64_t)nblk * kThreads) {
        const double n = c[b];
        if (n > 1.0 && n == floor(n)) s += lgamma(n + 1.0);
    }
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) r += sh[w];
        partial[(int64_t)t * nblk + blockIdx.x] = r;
    }
}

__global__ void k_rows_sum(const double* __restrict__ partial, int nblk, double* __restrict__ out, int64_t T) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[t * nblk + b];
    out[t] = s;
}

// Compatibility morph: out[r][b] = sum_c V[row(c, r)][b] * w_c in the reference's corner order
// with separate multiply and add (scipy _evaluate_linear: `value = value + term`), i.e.
// bit-identical to the CPU path.  rows of `src` have stride Bp, rows of `out` stride B.
__global__ __launch_bounds__(kThreads) void k_morph_store(const double* __restrict__ src,
                                                          const int64_t* __restrict__ rowoff,  // [R][nc]
                                                          const double* __restrict__ w,        // [nc]
                                                          int nc, int64_t B, double* __restrict__ out) {
    const int r = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    double v = 0.0;
    for (int c = 0; c < nc; ++c) {
        const double term = __dmul_rn(src[rowoff[(int64_t)r * nc + c] + b], w[c]);
        v = __dadd_rn(v, term);
    }
    out[(int64_t)r * B + b] = v;
}

// sum over bins of one padded row -> out[row]  (used for the Beeston-Barlow N table)
__global__ __launch_bounds__(kThreads) void k_row_total(const double* __restrict__ rows, int64_t B, int64_t Bp,
                                                        double* __restrict__ out) {
    const double* __restrict__ r = rows + (int64_t)blockIdx.x * Bp;
    double s = 0.0;
    for (int64_t b = threadIdx.x; b < B; b += kThreads) s += r[b];
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) t += sh[w];
        out[blockIdx.x] = t;
    }
}

// full_output with Beeston-Barlow (likelihood.py:634-658) on already-morphed templates:
// aw[b] = A_b * w_b and per-block partial sums of it.
__global__ __launch_bounds__(kThreads) void k_bb_full(const double* __restrict__ ps_m, const double* __restrict__ a_row,
                                                      const double* __restrict__ counts_row,
                                                      const double* __restrict__ mus, int S, int src, double p_cal,
                                                      double Ntot, int64_t B, double* __restrict__ aw,
                                                      double* __restrict__ partial) {
    double s = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < B; b += (int64_t)gridDim.x * kThreads) {
        double U = 0.0;
        for (int k = 0; k < S; ++k) {
            const double e = __dmul_rn(ps_m[(int64_t)k * B + b], k == src ? 0.0 : mus[k]);
            U = k == 0 ? e : __dadd_rn(U, e);
        }
        const double ab = a_row[b];
        const double w = ps_m[(int64_t)src * B + b] / ab * Ntot;
        double r1, r2;
        bb_roots(ab, w * p_cal, U, counts_row[b], r1, r2);
        const double A = (U == 0.0) ? (counts_row[b] + ab) / (1.0 + p_cal) : r2;
        const double v = A * w;
        aw[b] = v;
        s += v;
    }
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) t += sh[w];
        partial[blockIdx.x] = t;
    }
}

__global__ void k_bb_normalise(const double* __restrict__ aw, const double* __restrict__ tot, int64_t B,
                               double* __restrict__ row) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) row[b] = aw[b] / tot[0];
}

// per padded row: sum over bins, minimum, and a "has non-finite" flag -> out[row*3 + {0,1,2}]
__global__ __launch_bounds__(kThreads) void k_row_stats(const double* __restrict__ rows, int64_t B, int64_t Bp,
                                                        double* __restrict__ out) {
    const double* __restrict__ r = rows + (int64_t)blockIdx.x * Bp;
    double s = 0.0, mn = __builtin_inf(), bad = 0.0;
    for (int64_t b = threadIdx.x; b < B; b += kThreads) {
        const double v = r[b];
        s += v;
        mn = fmin(mn, v);
        if (!(fabs(v) < __builtin_inf())) bad = 1.0;
    }
    __shared__ double sh[3][kThreads / 64];
    s = wave_sum(s);
    bad = wave_sum(bad);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = mn; sh[2][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0][0], m = sh[1][0], f = sh[2][0];
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[0][w]; m = fmin(m, sh[1][w]); f += sh[2][w]; }
        out[(int64_t)blockIdx.x * 3 + 0] = t;
        out[(int64_t)blockIdx.x * 3 + 1] = m;
        out[(int64_t)blockIdx.x * 3 + 2] = f;
    }
}

// ---- non-empty-bin lists (CSR) of the datasets, built in bin order (deterministic) -----------
constexpr int kNzPerThread = 8;
constexpr int kNzChunk = kThreads * kNzPerThread;  // 2048 bins per block

__device__ __forceinline__ bool is_nz(double n) { return n != 0.0; }  // true for nan as well

__global__ __launch_bounds__(kThreads) void k_nz_count(const double* __restrict__ counts, int64_t B, int64_t Bp,
                                                       int32_t* __restrict__ cnt, int nchunks) {
    const double* __restrict__ c = counts + (int64_t)blockIdx.y * Bp;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    int k = 0;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j)
        if (b0 + j < B && is_nz(c[b0 + j])) ++k;
    __shared__ int sh[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(int64_t)blockIdx.y * nchunks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(kThreads) void k_nz_scatter(const double* __restrict__ counts, int64_t B, int64_t Bp,
                                                         const int64_t* __restrict__ chunk_off, int nchunks,
                                                         int32_t* __restrict__ nz_idx, double* __restrict__ nz_n) {
    const double* __restrict__ c = counts + (int64_t)blockIdx.y * Bp;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    double v[kNzPerThread];
    int k = 0;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j) {
        v[j] = (b0 + j < B) ? c[b0 + j] : 0.0;
        if (is_nz(v[j])) ++k;
    }
    // exclusive prefix of k over the block, in thread order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = k;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __shared__ int sh[kThreads / 64];
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    int64_t pos = chunk_off[(int64_t)blockIdx.y * nchunks + blockIdx.x] + base + incl - k;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j)
        if (is_nz(v[j])) {
            nz_idx[pos] = (int32_t)(b0 + j);
            nz_n[pos] = v[j];
            ++pos;
        }
}

// compacted templates of one dataset: out[row][j] = rows[row][idx[j]] (0 beyond nnz)
__global__ __launch_bounds__(kThreads) void k_gather_rows(const double* __restrict__ rows, int64_t Bp,
                                                          const int32_t* __restrict__ idx, int64_t nnz, int64_t np,
                                                          double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (j >= np) return;
    const int64_t row = blockIdx.y;
    out[row * np + j] = j < nnz ? rows[row * Bp + idx[j]] : 0.0;
}

__global__ void k_pad_copy(const double* __restrict__ src, int64_t n, int64_t np, double* __restrict__ dst) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < np) dst[j] = j < n ? src[j] : 0.0;
}

// toy-MC, CSR form: for dataset t: sum_j xlogy(n_j, mu[idx_j]); one block per dataset
__global__ __launch_bounds__(kThreads) void k_dataset_dot_csr(const int32_t* __restrict__ nz_idx,
                                                              const double* __restrict__ nz_n,
                                                              const int64_t* __restrict__ nz_off,
                                                              const double* __restrict__ logmu, int64_t t0,
                                                              double* __restrict__ partial) {
    const int64_t t = t0 + blockIdx.x;
    const int64_t lo = nz_off[t], hi = nz_off[t + 1];
    double s = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += kThreads) {
        const double n = nz_n[j];
        double term = n * logmu[nz_idx[j]];
        if (n != n) term = __builtin_nan("");
        else if (n < 0.0 || n != floor(n)) term = -__builtin_inf();
        s += term;
    }
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) r += sh[w];
        partial[blockIdx.x] = r;
    }
}

// ---- toy-MC generation on the device ---------------------------------------------------------
// n_{t,b} ~ Poisson(mu_b): the binned equivalent of Model.simulate (blueice/model.py:69-91: Poisson number of
// events per source, each drawn from the source's pdf) followed by set_data's binning (likelihood.py:603-609).
// Counter-based Philox4x32-10 keyed by the seed, counter = (bin, dataset, attempt): every (dataset, bin) draw
// is independent of launch geometry and can be regenerated, which is what lets the two-pass CSR build
// (count, then scatter) see the same numbers twice.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {  // uniform on [0, 1) with 53 random bits
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ double poisson_draw(double lam, uint64_t seed, int64_t t, int64_t b) {
    if (!(lam > 0.0)) return 0.0;  // mu = 0 (or invalid) -> no events
    uint32_t r[4];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (lam < 10.0) {
        // inversion by sequential search (one uniform)
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)t, (uint32_t)(t >> 32) & 0xFFFFu, k0, k1, r);
        const double u = u53(r[0], r[1]);
        double p = exp(-lam), F = p;
        double n = 0.0;
        while (u > F && n < 1000.0) {
            n += 1.0;
            p *= lam / n;
            F += p;
        }
        return n;
    }
    // PTRS, Hoermann (1993): transformed rejection with squeeze, as in numpy's random_poisson_ptrs
    const double slam = sqrt(lam), loglam = log(lam);
    const double bb = 0.931 + 2.53 * slam, aa = -0.059 + 0.02483 * bb;
    const double invalpha = 1.1239 + 1.1328 / (bb - 3.4), vr = 0.9277 - 3.6224 / (bb - 2.0);
    for (uint32_t attempt = 0; attempt < 4096u; ++attempt) {
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)t, ((uint32_t)(t >> 32) & 0xFFFFu) | ((attempt + 1u) << 16), k0, k1, r);
        const double U = u53(r[0], r[1]) - 0.5, V = u53(r[2], r[3]);
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * aa / us + bb) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(aa / (us * us) + bb) <= -lam + k * loglam - lgamma(k + 1.0)) return k;
    }
    return floor(lam);  // unreachable in practice (acceptance > 0.9 per attempt)
}

__global__ __launch_bounds__(kThreads) void k_toy_count(const double* __restrict__ mu, int64_t B, uint64_t seed, int64_t t0,
                                                        int32_t* __restrict__ cnt, int nchunks) {
    const int64_t t = t0 + blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    int k = 0;
#pragma unroll 1
    for (int j = 0; j < kNzPerThread; ++j)
        if (b0 + j < B && poisson_draw(mu[b0 + j], seed, t, b0 + j) != 0.0) ++k;
    __shared__ int sh[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(int64_t)blockIdx.y * nchunks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(kThreads) void k_toy_scatter(const double* __restrict__ mu, int64_t B, uint64_t seed, int64_t t0,
                                                          const int64_t* __restrict__ chunk_off, int nchunks,
                                                          int32_t* __restrict__ nz_idx, double* __restrict__ nz_n,
                                                          double* __restrict__ lg_partial) {
    const int64_t t = t0 + blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    double v[kNzPerThread];
    int k = 0;
    double lg = 0.0;
#pragma unroll 1
    for (int j = 0; j < kNzPerThread; ++j) {
        v[j] = (b0 + j < B) ? poisson_draw(mu[b0 + j], seed, t, b0 + j) : 0.0;
        if (v[j] != 0.0) { ++k; if (v[j] > 1.0) lg += lgamma(v[j] + 1.0); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = k;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int q = __shfl_up(incl, off, 64);
        if (lane >= off) incl += q;
    }
    __shared__ int sh[kThreads / 64];
    __shared__ double shl[kThreads / 64];
    lg = wave_sum(lg);
    if (lane == 63) sh[wave] = incl;
    if (lane == 0) shl[wave] = lg;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    int64_t pos = chunk_off[(int64_t)blockIdx.y * nchunks + blockIdx.x] + base + incl - k;
#pragma unroll
    for (int j = 0; j < kNzPerThread; ++j)
        if (v[j] != 0.0) {
            nz_idx[pos] = (int32_t)(b0 + j);
            nz_n[pos] = v[j];
            ++pos;
        }
    if (threadIdx.x == 0) lg_partial[(int64_t)blockIdx.y * nchunks + blockIdx.x] = shl[0] + shl[1] + shl[2] + shl[3];
}

// densify one dataset from its non-empty-bin list
__global__ void k_csr_to_dense(const int32_t* __restrict__ idx, const double* __restrict__ n, int64_t nnz,
                               double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) out[idx[j]] = n[j];
}

// ---- toy-MC form: one parameter point, many datasets --------------------------------------
// pass 1: mu_b -> logmu[b] (log mu, or -inf for mu == 0, or nan for invalid mu), partial sum mu
__global__ __launch_bounds__(kThreads) void k_morph_logmu(LaunchArgs a, double* __restrict__ logmu, int store_mu) {
    const int64_t* __restrict__ rowoff = a.rowoff;
    const double* __restrict__ coef = a.coef;
    double sum = 0.0;
    unsigned bad = 0u;
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double m0 = 0.0, m1 = 0.0;
#pragma unroll 8
        for (int k = 0; k < a.n0; ++k) {
            const double2 v = *reinterpret_cast<const double2*>(a.ps + rowoff[k] + bin0);
            const double c = coef[k];
            m0 = fma(c, v.x, m0);
            m1 = fma(c, v.y, m1);
        }
        double2 l;
        if (store_mu) {  // toy generation wants the expectation itself
            l.x = m0;
            l.y = m1;
        } else {
            l.x = (m0 >= 0.0) ? log(m0) : __builtin_nan("");
            l.y = (m1 >= 0.0) ? log(m1) : __builtin_nan("");
        }
        if (!(m0 >= 0.0) || !(m1 >= 0.0)) bad = 1u;
        *reinterpret_cast<double2*>(logmu + bin0) = l;
        sum += m0 + m1;
    }
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    sum = wave_sum(sum);
    bad = wave_or(bad);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = sum; shf[threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        unsigned f = shf[0];
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[w]; f |= shf[w]; }
        a.partial[blockIdx.x] = t;
        a.pflags[blockIdx.x] = f;
    }
}

// pass 2: for dataset t: sum_b xlogy(n_tb, mu_b) ; blockIdx.y = dataset, x strides tiles
__global__ __launch_bounds__(kThreads) void k_dataset_dot(const double* __restrict__ counts,
                                                          const double* __restrict__ logmu, int64_t Bp, int n_tiles,
                                                          int64_t t0, double* __restrict__ partial) {
    const double* __restrict__ c = counts + (t0 + blockIdx.y) * Bp;
    double s = 0.0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        const double2 n = *reinterpret_cast<const double2*>(c + bin0);
        const double2 l = *reinterpret_cast<const double2*>(logmu + bin0);
        double t0v = (n.x > 0.0) ? n.x * l.x : 0.0;
        double t1v = (n.y > 0.0) ? n.y * l.y : 0.0;
        if (n.x != n.x) t0v = __builtin_nan("");
        else if (n.x < 0.0 || n.x != floor(n.x)) t0v = -__builtin_inf();
        if (n.y != n.y) t1v = __builtin_nan("");
        else if (n.y < 0.0 || n.y != floor(n.y)) t1v = -__builtin_inf();
        s += t0v + t1v;
    }
    __shared__ double sh[kThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int w = 1; w < kThreads / 64; ++w) t += sh[w];
        partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// out[t] = sum_blocks partial[t][:] - summu - lgsum[t0 + t]   (nan if any mu invalid)
__global__ void k_dataset_finish(const double* __restrict__ partial, int nbx, const double* __restrict__ mu_partial,
                                 const unsigned* __restrict__ mu_flags, int nmu, const double* __restrict__ lgsum,
                                 int64_t t0, int64_t n, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double s = 0.0;
    for (int b = 0; b < nbx; ++b) s += partial[t * nbx + b];
    double m = 0.0;
    unsigned f = 0u;
    for (int b = 0; b < nmu; ++b) { m += mu_partial[b]; f |= mu_flags[b]; }
    double r = (s - m) - lgsum[t0 + t];
    if (f) r = __builtin_nan("");
    out[t] = r;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

struct bi_plan {
    int64_t P = 0;
    struct Class {
        int G = 0;
        int64_t n_items = 0;
        int nbx = 0;
        DevBuf rowoff, coef, aux, item_cnt, item_tiles, perm, slot_lg, partial, pflags;
    };
    std::vector<Class> classes;
    DevBuf bad_idx;            // points answered on the host side with -inf
    int64_t n_bad = 0;
    DevBuf out, status;        // internal result buffers [P]
    std::vector<int32_t> h_status;
    int64_t epoch = 0;         // ctx->epoch at creation: a plan dies with the model / data it was made for
    bool no_reuse = false;     // no anchor model is touched by two items of the plan
    bool sparse = false;       // rows / counts refer to the compacted (non-empty-bin) copies
    int64_t bytes = 0;         // algorithmic HBM bytes per run
    int64_t launches = 0;
};

struct bi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    std::string err;

    // model
    bool model_ready = false, model_open = false;
    int d = 0, S = 0;
    int64_t B = 0, Bp = 0, A = 0;
    std::vector<int> n_anchor;
    std::vector<std::vector<double>> grid;
    std::vector<int64_t> astride;  // anchor-index stride per axis
    std::vector<int> eff_axes;     // axes with >= 2 anchors
    int bb_source = -1;
    std::vector<int32_t> allow_neg;
    DevBuf ps, nm, nm_tot;
    std::vector<double> h_mus;     // [A][S]
    std::vector<double> h_nm_tot;  // [A]
    std::vector<char> anchor_set;

    // data
    bool unbinned = false;      // extended unbinned likelihood: rows are pdf values at the events
    double outlier = 0.0;
    bool ps_finite = true;
    bool data_ready = false;
    bool dense_counts = false;  // counts [T][Bp] resident (false for device-generated toys: CSR lists only)
    int64_t T = 0;
    DevBuf counts, lgsum;
    std::vector<double> h_lgsum;

    // model statistics (for the sparse forms)
    std::vector<double> h_rowsum;  // [A*S] sum over bins of every ps row
    bool ps_nonneg = false;        // every ps entry is finite and >= 0

    // sparse forms of the data: CSR lists of the non-empty bins, and per-dataset compacted templates
    bool csr_ready = false, compact_ready = false;
    DevBuf nz_idx, nz_n, nz_off, ps_c, cnt_c;
    std::vector<int64_t> h_nz_off;            // [T+1]
    std::vector<int64_t> h_c_off, h_cnt_off;  // [T] element offsets into ps_c / cnt_c
    std::vector<int64_t> h_c_np;              // [T] padded non-empty bins per dataset
    std::vector<double> h_Tz;                 // [T][A*S] sum of every ps row over the EMPTY bins of the dataset

    // persistent single-point slot (the lf(**kw) call shape): no allocation, one H2D, one D2H per call
    DevBuf slot_dev, slot_partial, slot_pflags;
    void* slot_host = nullptr;  // pinned staging: descriptors in, {ll, status} out
    size_t slot_host_bytes = 0;

    // scratch
    DevBuf scratch, scratch2, logmu;

    int64_t epoch = 0;  // bumped by every model / data upload

    // tunables
    int64_t blocks_per_cu = 8;
    int64_t max_group = kMaxG;
    int64_t nt_loads = 2;                        // nontemporal template loads: 0 never, 1 always, 2 when no reuse
    int64_t sparse = 1;                          // use the sparse forms when they are exactly equivalent
    int64_t compact_budget = (int64_t)16 << 30;  // bytes of HBM the compacted templates may take

    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
};

namespace {

int fail(bi_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), BI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int dev_alloc(bi_ctx* c, DevBuf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return BI_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, BI_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    b.bytes = bytes;
    return BI_OK;
}

void dev_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

template <class T>
int dev_upload(bi_ctx* c, DevBuf& b, const std::vector<T>& h) {
    int rc = dev_alloc(c, b, h.size() * sizeof(T));
    if (rc) return rc;
    if (!h.empty()) HIP_TRY(c, hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return BI_OK;
}

void free_plan_buffers(bi_plan* p) {
    for (auto& k : p->classes) {
        dev_free(k.rowoff); dev_free(k.coef); dev_free(k.aux); dev_free(k.item_cnt); dev_free(k.item_tiles);
        dev_free(k.perm); dev_free(k.slot_lg); dev_free(k.partial); dev_free(k.pflags);
    }
    dev_free(p->bad_idx); dev_free(p->out); dev_free(p->status);
}

// scipy find_indices semantics on one axis (oracle/blueice_oracle.py:find_cell)
inline void find_cell(const std::vector<double>& g, double z, int& k, double& t) {
    const int n = (int)g.size();
    if (n == 1) { k = 0; t = 0.0; return; }
    if (z == g[n - 1]) {
        k = n - 2;
    } else {
        k = (int)(std::upper_bound(g.begin(), g.end(), z) - g.begin()) - 1;
        k = std::min(std::max(k, 0), n - 2);
    }
    const double denom = g[k + 1] - g[k];
    t = (z - g[k]) / denom;
}

struct PointGeom {
    int64_t cell_anchor;          // linear anchor index of the lower corner
    std::vector<double> w;        // [2^deff] corner weights, reference order
    double t[kMaxDim];            // per axis: normalised distance in the cell
    double inv_delta[kMaxDim];    // per axis: 1 / (g[k+1] - g[k])  (0 for single-anchor axes)
};

// corner c (bit i from the most significant = effective axis 0) -> anchor offset
inline int64_t corner_offset(const bi_ctx* c, int corner) {
    const int de = (int)c->eff_axes.size();
    int64_t off = 0;
    for (int i = 0; i < de; ++i)
        if ((corner >> (de - 1 - i)) & 1) off += c->astride[c->eff_axes[i]];
    return off;
}

// returns false when z is outside the anchor box (or nan): likelihood.py:345-347
bool point_geometry(const bi_ctx* c, const double* z, PointGeom& g) {
    for (int i = 0; i < c->d; ++i) {
        const auto& gr = c->grid[i];
        if (!(gr.front() <= z[i] && z[i] <= gr.back())) return false;
    }
    const int de = (int)c->eff_axes.size();
    int kk[kMaxDim];
    double tt[kMaxDim];
    int64_t base = 0;
    for (int i = 0; i < c->d; ++i) {
        int k; double t;
        find_cell(c->grid[i], z[i], k, t);
        base += (int64_t)k * c->astride[i];
        kk[i] = k; tt[i] = t;
        g.t[i] = t;
        g.inv_delta[i] = c->grid[i].size() > 1 ? 1.0 / (c->grid[i][(size_t)k + 1] - c->grid[i][(size_t)k]) : 0.0;
    }
    (void)kk;
    g.cell_anchor = base;
    const int nc = 1 << de;
    g.w.assign(nc, 1.0);
    for (int corner = 0; corner < nc; ++corner) {
        double w = 1.0;
        for (int i = 0; i < de; ++i) {
            const double t = tt[c->eff_axes[i]];
            const double wi = ((corner >> (de - 1 - i)) & 1) ? t : (1 - t);
            w = w * wi;
        }
        g.w[corner] = w;
    }
    return true;
}

// mus_interpolator(z): value = value + V*w per corner, left to right from 0.0
void interp_mus(const bi_ctx* c, const PointGeom& g, double* mus) {
    const int nc = (int)g.w.size();
    for (int s = 0; s < c->S; ++s) {
        double v = 0.0;
        for (int corner = 0; corner < nc; ++corner) {
            const int64_t a = g.cell_anchor + corner_offset(c, corner);
            const double term = c->h_mus[a * c->S + s] * g.w[corner];
            v = v + term;
        }
        mus[s] = v;
    }
}

// likelihood.py:397-415
bool rates_physical(const bi_ctx* c, const double* mus) {
    const double inf = std::numeric_limits<double>::infinity();
    bool any_allowed = false;
    for (int s = 0; s < c->S; ++s) any_allowed |= (c->allow_neg[s] != 0);
    if (!any_allowed) {
        for (int s = 0; s < c->S; ++s)
            if (!(mus[s] >= 0 && mus[s] < inf)) return false;
        return true;
    }
    bool any_fin = false;
    double tot = 0;
    for (int s = 0; s < c->S; ++s) { any_fin |= (mus[s] < inf); tot += mus[s]; }
    if (!any_fin || tot < 0) return false;
    for (int s = 0; s < c->S; ++s)
        if (!(0 <= mus[s]) && !c->allow_neg[s]) return false;
    return true;
}

int pick_class(int n, int maxg) {
    int g = 1;
    while (g < n && g < maxg) g <<= 1;
    return g;
}

struct EventScope {
    bi_ctx* c;
    size_t idx = (size_t)-1;
    explicit EventScope(bi_ctx* ctx) : c(ctx) {
        if (!c->profiling) return;
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            c->ev_pool.emplace_back(a, b);
        }
        idx = c->ev_used++;
        (void)hipEventRecord(c->ev_pool[idx].first, c->stream);
    }
    ~EventScope() {
        if (idx != (size_t)-1) (void)hipEventRecord(c->ev_pool[idx].second, c->stream);
    }
};

template <int G>
void launch_morph(bi_ctx* c, const LaunchArgs& a, dim3 grid, bool bb, bool nt) {
    if (c->unbinned) {
        if (nt) hipLaunchKernelGGL((k_morph_reduce<G, false, true, 2>), grid, dim3(kThreads), 0, c->stream, a);
        else hipLaunchKernelGGL((k_morph_reduce<G, false, false, 2>), grid, dim3(kThreads), 0, c->stream, a);
        return;
    }
    if (bb && nt) hipLaunchKernelGGL((k_morph_reduce<G, true, true>), grid, dim3(kThreads), 0, c->stream, a);
    else if (bb) hipLaunchKernelGGL((k_morph_reduce<G, true, false>), grid, dim3(kThreads), 0, c->stream, a);
    else if (nt) hipLaunchKernelGGL((k_morph_reduce<G, false, true>), grid, dim3(kThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((k_morph_reduce<G, false, false>), grid, dim3(kThreads), 0, c->stream, a);
}

void launch_morph_grad(bi_ctx* c, int G, const LaunchArgs& a, dim3 grid, bool nt) {
    EventScope ev(c);
#define BI_GRAD_CASE(GG)                                                                                          \
    case GG:                                                                                                      \
        if (nt) hipLaunchKernelGGL((k_morph_reduce<GG, false, true, 1>), grid, dim3(kThreads), 0, c->stream, a); \
        else hipLaunchKernelGGL((k_morph_reduce<GG, false, false, 1>), grid, dim3(kThreads), 0, c->stream, a);   \
        break;
    switch (G) {
        BI_GRAD_CASE(2)
        BI_GRAD_CASE(4)
        BI_GRAD_CASE(8)
        default:
            BI_GRAD_CASE(16)
    }
#undef BI_GRAD_CASE
}

// nt: the launch streams its template rows exactly once (no two items touch the same anchor), so the loads
// carry the nontemporal hint: +8 % HBM rate on gfx950; with shared rows the default policy (L2 / MALL) wins.
void launch_morph_g(bi_ctx* c, int G, const LaunchArgs& a, dim3 grid, bool bb, bool nt) {
    EventScope ev(c);
    switch (G) {
        case 1: launch_morph<1>(c, a, grid, bb, nt); break;
        case 2: launch_morph<2>(c, a, grid, bb, nt); break;
        case 4: launch_morph<4>(c, a, grid, bb, nt); break;
        case 8: launch_morph<8>(c, a, grid, bb, nt); break;
        default: launch_morph<16>(c, a, grid, bb, nt); break;
    }
}

int check_ready(bi_ctx* c, bool need_data) {
    if (!c) return BI_ERR_INVALID;
    if (!c->model_ready) return fail(c, BI_ERR_STATE, "no model uploaded (prepare() first)");
    if (need_data && !c->data_ready) return fail(c, BI_ERR_STATE, "no data uploaded (set_data() first)");
    return BI_OK;
}

int n_tiles_of(const bi_ctx* c) { return (int)(c->Bp / kTile); }

// per-dataset compacted copies of all template rows over the non-empty bins (needs the CSR lists)
int build_compact_templates(bi_ctx* c) {
    c->compact_ready = false;
    const int64_t T = c->T, Bp = c->Bp;
    int rc;
    hipError_t e;
    if (!c->ps_nonneg || c->bb_source >= 0) return BI_OK;
    const int64_t rows = c->A * c->S;
    c->h_c_np.assign((size_t)T, 0);
    c->h_c_off.assign((size_t)T, 0);
    c->h_cnt_off.assign((size_t)T, 0);
    int64_t tot_ps = 0, tot_cnt = 0;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t nnz = c->h_nz_off[(size_t)t + 1] - c->h_nz_off[(size_t)t];
        const int64_t np = std::max<int64_t>(kTile, (nnz + kTile - 1) / kTile * kTile);
        c->h_c_np[(size_t)t] = np;
        c->h_c_off[(size_t)t] = tot_ps;
        c->h_cnt_off[(size_t)t] = tot_cnt;
        tot_ps += rows * np;
        tot_cnt += np;
    }
    if ((tot_ps + tot_cnt) * (int64_t)sizeof(double) > c->compact_budget) return BI_OK;
    if ((rc = dev_alloc(c, c->ps_c, (size_t)tot_ps * sizeof(double))) || (rc = dev_alloc(c, c->cnt_c, (size_t)tot_cnt * sizeof(double))) ||
        (rc = dev_alloc(c, c->scratch, (size_t)rows * sizeof(double))))
        return rc;
    c->h_Tz.assign((size_t)T * rows, 0.0);
    std::vector<double> tnz((size_t)rows);
    for (int64_t t = 0; t < T; ++t) {
        const int64_t lo = c->h_nz_off[(size_t)t], nnz = c->h_nz_off[(size_t)t + 1] - lo, np = c->h_c_np[(size_t)t];
        double* dst = (double*)c->ps_c.p + c->h_c_off[(size_t)t];
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((np + kThreads - 1) / kThreads), (unsigned)rows), dim3(kThreads), 0,
                           c->stream, (const double*)c->ps.p, Bp, (const int32_t*)c->nz_idx.p + lo, nnz, np, dst);
        hipLaunchKernelGGL(k_pad_copy, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)c->nz_n.p + lo, nnz, np, (double*)c->cnt_c.p + c->h_cnt_off[(size_t)t]);
        hipLaunchKernelGGL(k_row_total, dim3((unsigned)rows), dim3(kThreads), 0, c->stream, (const double*)dst, np, np,
                           (double*)c->scratch.p);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(tnz.data(), c->scratch.p, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, BI_ERR_HIP, "template compaction: %s", hipGetErrorString(e));
        for (int64_t r = 0; r < rows; ++r) c->h_Tz[(size_t)(t * rows + r)] = c->h_rowsum[(size_t)r] - tnz[(size_t)r];
    }
    c->compact_ready = true;
    return BI_OK;
}

// CSR lists of the non-empty bins of every dataset (always, unless the data are dense), and -- when the
// templates are non-negative and the budget allows -- per-dataset compacted copies of all template rows,
// so that an evaluation only touches non-empty bins:
//   sum_b [n log mu - mu - lgamma(n+1)] = sum_{b: n_b != 0} [n log mu - mu] - sum_k coef_k Tz_k - sum lgamma
// with Tz_k = sum of row k over the EMPTY bins.  Exact (to rounding) because mu_b >= 0 is then guaranteed,
// so the only per-bin terms that are not linear in the templates are those of the non-empty bins.
int build_sparse_forms(bi_ctx* c) {
    c->csr_ready = c->compact_ready = false;
    const int64_t T = c->T, B = c->B, Bp = c->Bp;
    const int nchunks = (int)((B + kNzChunk - 1) / kNzChunk);
    int rc;
    DevBuf d_cnt, d_off;
    auto cleanup = [&]() { dev_free(d_cnt); dev_free(d_off); };
    if ((rc = dev_alloc(c, d_cnt, (size_t)T * nchunks * sizeof(int32_t)))) return rc;
    const int64_t tchunk = 32768;
    for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
        const int64_t n = std::min(tchunk, T - t0);
        hipLaunchKernelGGL(k_nz_count, dim3((unsigned)nchunks, (unsigned)n), dim3(kThreads), 0, c->stream,
                           (const double*)c->counts.p + t0 * Bp, B, Bp, (int32_t*)d_cnt.p + t0 * nchunks, nchunks);
    }
    std::vector<int32_t> h_cnt((size_t)T * nchunks);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt.data(), d_cnt.p, h_cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return fail(c, BI_ERR_HIP, "non-empty-bin count: %s", hipGetErrorString(e)); }
    std::vector<int64_t> h_off(h_cnt.size());
    c->h_nz_off.assign((size_t)T + 1, 0);
    int64_t run = 0;
    for (int64_t t = 0; t < T; ++t) {
        c->h_nz_off[(size_t)t] = run;
        for (int k = 0; k < nchunks; ++k) { h_off[(size_t)t * nchunks + k] = run; run += h_cnt[(size_t)t * nchunks + k]; }
    }
    c->h_nz_off[(size_t)T] = run;
    if (c->sparse == 0 || (c->sparse == 1 && run > T * B / 4)) { cleanup(); return BI_OK; }  // dense data: dense forms
    if ((rc = dev_upload(c, d_off, h_off)) || (rc = dev_alloc(c, c->nz_idx, (size_t)std::max<int64_t>(run, 1) * sizeof(int32_t))) ||
        (rc = dev_alloc(c, c->nz_n, (size_t)std::max<int64_t>(run, 1) * sizeof(double))) || (rc = dev_upload(c, c->nz_off, c->h_nz_off))) {
        cleanup();
        return rc;
    }
    for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
        const int64_t n = std::min(tchunk, T - t0);
        hipLaunchKernelGGL(k_nz_scatter, dim3((unsigned)nchunks, (unsigned)n), dim3(kThreads), 0, c->stream,
                           (const double*)c->counts.p + t0 * Bp, B, Bp, (const int64_t*)d_off.p + t0 * nchunks, nchunks,
                           (int32_t*)c->nz_idx.p, (double*)c->nz_n.p);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "non-empty-bin scatter: %s", hipGetErrorString(e));
    c->csr_ready = true;

    return build_compact_templates(c);
}



// One point, synchronous: the call shape of `lf(**kwargs)` inside a minimizer (inference.py:111-122 makes
// ~500 of them per fit).  Same kernels as the batched path, but the descriptors live in a persistent
// device slot fed from pinned memory: one small H2D, two launches, one 16-byte D2H, one sync.
int eval_single(bi_ctx* c, const double* z, const double* rate_scale, int64_t ds, double* out, int32_t* status) {
    const int S = c->S;
    const double ninf = -std::numeric_limits<double>::infinity();
    if (ds < 0 || ds >= c->T) { *out = ninf; if (status) *status = BI_ST_BAD_DATASET; return BI_OK; }
    PointGeom g;
    if (!point_geometry(c, z, g)) { *out = ninf; if (status) *status = BI_ST_OUT_OF_BOUNDS; return BI_OK; }
    double r[64];
    std::vector<double> rbig;
    double* rates = r;
    if (S > 64) { rbig.resize((size_t)S); rates = rbig.data(); }
    interp_mus(c, g, rates);
    if (rate_scale) for (int s = 0; s < S; ++s) rates[s] *= rate_scale[s];
    if (!rates_physical(c, rates)) { *out = ninf; if (status) *status = BI_ST_UNPHYSICAL; return BI_OK; }

    const bool bb = c->bb_source >= 0;
    const int nc = (int)g.w.size();
    const int n0 = bb ? nc * (S - 1) : nc * S, n1 = bb ? nc : 0, n2 = bb ? nc : 0, NS = n0 + n1 + n2;
    bool any_neg = false;
    for (int q = 0; q < S; ++q) any_neg |= (c->allow_neg[(size_t)q] != 0);
    const bool sparse = c->sparse && c->compact_ready && !bb && !any_neg && !c->unbinned;
    if (!sparse && !c->dense_counts) return fail(c, BI_ERR_STATE, "dataset counts are not resident in dense form");
    const int64_t row_stride = sparse ? c->h_c_np[(size_t)ds] : c->Bp;
    const int64_t row_base = sparse ? c->h_c_off[(size_t)ds] : 0;
    const int tiles = (int)(row_stride / kTile);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nbx = (int)std::min<int64_t>(tiles, slots);

    // slot layout (8-byte units): rowoff[NS] coef[NS] aux[2] cnt_off tiles perm slot_lg | result {ll, status}
    const size_t n_words = (size_t)NS * 2 + 2 + 4 + 2;
    const size_t bytes = n_words * 8;
    int rc;
    if (c->slot_host_bytes < bytes) {
        if (c->slot_host) (void)hipHostFree(c->slot_host);
        c->slot_host = nullptr;
        HIP_TRY(c, hipHostMalloc(&c->slot_host, bytes * 2, hipHostMallocDefault));
        c->slot_host_bytes = bytes * 2;
    }
    if ((rc = dev_alloc(c, c->slot_dev, bytes)) || (rc = dev_alloc(c, c->slot_partial, (size_t)slots * sizeof(double))) ||
        (rc = dev_alloc(c, c->slot_pflags, (size_t)slots * sizeof(unsigned))))
        return rc;
    int64_t* w64 = (int64_t*)c->slot_host;
    double* wd = (double*)c->slot_host;
    int64_t* rowoff = w64;
    double* coef = wd + NS;
    double* aux = wd + 2 * NS;
    int k = 0;
    double zsum = 0.0;
    const int64_t n_rows = c->A * S;
    for (int corner = 0; corner < nc; ++corner) {
        const int64_t a = g.cell_anchor + corner_offset(c, corner);
        for (int s = 0; s < S; ++s) {
            if (bb && s == c->bb_source) continue;
            rowoff[k] = row_base + (a * S + s) * row_stride;
            coef[k] = g.w[(size_t)corner] * rates[s];
            if (sparse) zsum += coef[k] * c->h_Tz[(size_t)(ds * n_rows + a * S + s)];
            ++k;
        }
    }
    aux[0] = 1.0; aux[1] = 1.0;
    if (bb) {
        double Ntot = 0.0;
        for (int corner = 0; corner < nc; ++corner) {
            const int64_t a = g.cell_anchor + corner_offset(c, corner);
            rowoff[n0 + corner] = (a * S + c->bb_source) * c->Bp;
            coef[n0 + corner] = g.w[(size_t)corner];
            rowoff[n0 + n1 + corner] = a * c->Bp;
            coef[n0 + n1 + corner] = g.w[(size_t)corner];
            const double term = c->h_nm_tot[(size_t)a] * g.w[(size_t)corner];
            Ntot = Ntot + term;
        }
        aux[0] = rates[c->bb_source] / Ntot;
        aux[1] = Ntot;
    }
    const size_t o = (size_t)2 * NS + 2;
    w64[o + 0] = sparse ? c->h_cnt_off[(size_t)ds] : ds * c->Bp;   // cnt_off
    ((int32_t*)(w64 + o + 1))[0] = tiles;                            // tiles (+ pad)
    ((int32_t*)(w64 + o + 1))[1] = 0;
    w64[o + 2] = 0;                                                  // perm -> out[0]
    wd[o + 3] = c->h_lgsum[(size_t)ds] + zsum;                       // slot_lg
    if (c->unbinned) {
        double rsum = 0.0;
        for (int s = 0; s < S; ++s) rsum += rates[s];
        wd[o + 3] = rsum;
    }
    // the result {ll, status} is written by k_finish straight into the pinned host block (second half)
    char* res = (char*)c->slot_host + bytes;
    *(double*)res = 0.0;
    *(int64_t*)(res + 8) = 0;

    char* dev = (char*)c->slot_dev.p;
    HIP_TRY(c, hipMemcpyAsync(dev, c->slot_host, bytes, hipMemcpyHostToDevice, c->stream));
    LaunchArgs a{};
    a.ps = sparse ? (const double*)c->ps_c.p : (const double*)c->ps.p;
    a.nm = (const double*)c->nm.p;
    a.counts = sparse ? (const double*)c->cnt_c.p : (const double*)c->counts.p;
    a.rowoff = (const int64_t*)dev;
    a.coef = (const double*)(dev + (size_t)NS * 8);
    a.aux = (const double*)(dev + (size_t)NS * 16);
    a.item_cnt = (const int64_t*)(dev + (o + 0) * 8);
    a.item_tiles = (const int32_t*)(dev + (o + 1) * 8);
    a.partial = (double*)c->slot_partial.p;
    a.pflags = (unsigned*)c->slot_pflags.p;
    a.B = c->B; a.Bp = c->Bp; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.n_tiles = tiles;
    a.outlier = c->outlier;
    launch_morph_g(c, 1, a, dim3((unsigned)nbx, 1), bb, !sparse && c->nt_loads != 0);
    const int lanes = nbx > 64 ? kThreads : 64;
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(kThreads), 0, c->stream, (const double*)a.partial,
                       (const unsigned*)a.pflags, nbx, 1, lanes, (int64_t)1, (const int64_t*)(dev + (o + 2) * 8),
                       (const double*)(dev + (o + 3) * 8), (double*)res, (int32_t*)(res + 8));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *out = *(double*)res;
    if (status) *status = *(int32_t*)(res + 8);
    return BI_OK;
}

}  // namespace

extern "C" {

const char* bi_version(void) { return BI_VERSION; }

const char* bi_last_error(const bi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int bi_create(int device, bi_ctx** out) {
    if (!out) return fail(nullptr, BI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, BI_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, BI_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    bi_ctx* c = new bi_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&c->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(nullptr, BI_ERR_HIP, "device init failed: %s", hipGetErrorString(e));
        delete c;
        return BI_ERR_HIP;
    }
    *out = c;
    return BI_OK;
}

void bi_destroy(bi_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    dev_free(c->ps); dev_free(c->nm); dev_free(c->nm_tot); dev_free(c->counts); dev_free(c->lgsum);
    dev_free(c->scratch); dev_free(c->scratch2); dev_free(c->logmu);
    dev_free(c->slot_dev); dev_free(c->slot_partial); dev_free(c->slot_pflags);
    if (c->slot_host) (void)hipHostFree(c->slot_host);
    dev_free(c->nz_idx); dev_free(c->nz_n); dev_free(c->nz_off); dev_free(c->ps_c); dev_free(c->cnt_c);
    for (auto& ev : c->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int bi_device_info(bi_ctx* c, char* name, char* arch, int len, int* n_cu, int64_t* hbm_bytes) {
    if (!c) return BI_ERR_INVALID;
    if (name && len > 0) { strncpy(name, c->prop.name, len - 1); name[len - 1] = 0; }
    if (arch && len > 0) { strncpy(arch, c->prop.gcnArchName, len - 1); arch[len - 1] = 0; }
    if (n_cu) *n_cu = c->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)c->prop.totalGlobalMem;
    return BI_OK;
}

void* bi_stream(bi_ctx* c) { return c ? (void*)c->stream : nullptr; }

int bi_sync(bi_ctx* c) {
    if (!c) return BI_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BI_OK;
}

int bi_set_param(bi_ctx* c, const char* name, int64_t v) {
    if (!c || !name) return BI_ERR_INVALID;
    if (!strcmp(name, "blocks_per_cu")) { if (v < 1 || v > 32) return fail(c, BI_ERR_INVALID, "blocks_per_cu in [1,32]"); c->blocks_per_cu = v; return BI_OK; }
    if (!strcmp(name, "max_group")) {
        if (v < 1 || v > kMaxG || (v & (v - 1))) return fail(c, BI_ERR_INVALID, "max_group must be a power of two in [1,%d]", kMaxG);
        c->max_group = v;
        return BI_OK;
    }
    if (!strcmp(name, "sparse")) {
        if (v < 0 || v > 2) return fail(c, BI_ERR_INVALID, "sparse: 0 = off, 1 = auto, 2 = whenever exact");
        c->sparse = v;
        return BI_OK;
    }
    if (!strcmp(name, "compact_budget")) { c->compact_budget = v; return BI_OK; }
    if (!strcmp(name, "nt_loads")) {
        if (v < 0 || v > 2) return fail(c, BI_ERR_INVALID, "nt_loads: 0 = never, 1 = always, 2 = auto");
        c->nt_loads = v;
        return BI_OK;
    }
    return fail(c, BI_ERR_INVALID, "unknown parameter %s", name);
}

int64_t bi_get_param(bi_ctx* c, const char* name) {
    if (!c || !name) return -1;
    if (!strcmp(name, "blocks_per_cu")) return c->blocks_per_cu;
    if (!strcmp(name, "max_group")) return c->max_group;
    if (!strcmp(name, "tile_bins")) return kTile;
    if (!strcmp(name, "padded_bins")) return c->Bp;
    if (!strcmp(name, "sparse")) return c->sparse;
    if (!strcmp(name, "nt_loads")) return c->nt_loads;
    if (!strcmp(name, "compact_budget")) return c->compact_budget;
    if (!strcmp(name, "csr_ready")) return c->csr_ready ? 1 : 0;
    if (!strcmp(name, "compact_ready")) return c->compact_ready ? 1 : 0;
    if (!strcmp(name, "ps_nonneg")) return c->ps_nonneg ? 1 : 0;
    if (!strcmp(name, "nnz_total")) return c->csr_ready ? c->h_nz_off.back() : -1;
    return -1;
}

// ---- model ---------------------------------------------------------------------------------

int bi_model_begin(bi_ctx* c, int d, const int32_t* n_anchor, const double* anchor_z, int S, int64_t B,
                   int bb_source) {
    if (!c) return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (d < 0 || d > kMaxDim) return fail(c, BI_ERR_INVALID, "d=%d outside [0,%d]", d, kMaxDim);
    if (S < 1 || B < 0) return fail(c, BI_ERR_INVALID, "need S >= 1 and B >= 0 (got S=%d B=%lld)", S, (long long)B);
    if (bb_source < -1 || bb_source >= S) return fail(c, BI_ERR_INVALID, "bb_source %d outside [-1,%d)", bb_source, S);
    if (d > 0 && (!n_anchor || !anchor_z)) return fail(c, BI_ERR_INVALID, "anchor arrays are NULL");
    c->model_ready = false;
    c->data_ready = false;  // a new model invalidates the data (likelihood.py:253)
    ++c->epoch;
    c->d = d; c->S = S; c->B = B; c->bb_source = bb_source;
    c->Bp = std::max<int64_t>(kTile, (B + kTile - 1) / kTile * kTile);
    c->unbinned = false;
    c->n_anchor.assign(d, 0);
    c->grid.assign(d, {});
    c->A = 1;
    const double* zp = anchor_z;
    for (int i = 0; i < d; ++i) {
        if (n_anchor[i] < 1) return fail(c, BI_ERR_INVALID, "axis %d has %d anchors", i, n_anchor[i]);
        c->n_anchor[i] = n_anchor[i];
        c->grid[i].assign(zp, zp + n_anchor[i]);
        for (int j = 1; j < n_anchor[i]; ++j)
            if (!(c->grid[i][j] > c->grid[i][j - 1]))
                return fail(c, BI_ERR_INVALID, "anchor z values of axis %d are not strictly ascending", i);
        zp += n_anchor[i];
        c->A *= n_anchor[i];
    }
    c->astride.assign(d, 1);
    for (int i = d - 2; i >= 0; --i) c->astride[i] = c->astride[i + 1] * c->n_anchor[i + 1];
    c->eff_axes.clear();
    for (int i = 0; i < d; ++i)
        if (c->n_anchor[i] >= 2) c->eff_axes.push_back(i);
    c->allow_neg.assign(S, 0);
    c->h_mus.assign((size_t)c->A * S, 0.0);
    c->h_nm_tot.assign((size_t)c->A, 0.0);
    c->anchor_set.assign((size_t)c->A, 0);
    const size_t ps_bytes = (size_t)c->A * S * c->Bp * sizeof(double);
    int rc = dev_alloc(c, c->ps, ps_bytes);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->ps.p, 0, ps_bytes, c->stream));
    if (bb_source >= 0) {
        const size_t nm_bytes = (size_t)c->A * c->Bp * sizeof(double);
        if ((rc = dev_alloc(c, c->nm, nm_bytes))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->nm.p, 0, nm_bytes, c->stream));
        if ((rc = dev_alloc(c, c->nm_tot, (size_t)c->A * sizeof(double)))) return rc;
    }
    c->model_open = true;
    return BI_OK;
}

int bi_model_set_anchor(bi_ctx* c, int64_t ai, const double* ps, const double* mus, const double* nm_row) {
    if (!c || !c->model_open) return fail(c, BI_ERR_STATE, "bi_model_begin first");
    if (ai < 0 || ai >= c->A) return fail(c, BI_ERR_INVALID, "anchor index %lld outside [0,%lld)", (long long)ai, (long long)c->A);
    if (!ps || !mus) return fail(c, BI_ERR_INVALID, "ps / mus are NULL");
    if (c->bb_source >= 0 && !nm_row) return fail(c, BI_ERR_INVALID, "Beeston-Barlow model needs the n_model row");
    HIP_TRY(c, hipSetDevice(c->device));
    double* dst = (double*)c->ps.p + (size_t)ai * c->S * c->Bp;
    if (c->B > 0) HIP_TRY(c, hipMemcpy2DAsync(dst, c->Bp * sizeof(double), ps, c->B * sizeof(double), c->B * sizeof(double), c->S,
                                hipMemcpyHostToDevice, c->stream));
    for (int s = 0; s < c->S; ++s) c->h_mus[(size_t)ai * c->S + s] = mus[s];
    if (c->bb_source >= 0 && c->B > 0) {
        double* nd = (double*)c->nm.p + (size_t)ai * c->Bp;
        HIP_TRY(c, hipMemcpyAsync(nd, nm_row, c->B * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    // the host buffers are borrowed only for the duration of the call
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->anchor_set[(size_t)ai] = 1;
    return BI_OK;
}

int bi_model_end(bi_ctx* c) {
    if (!c || !c->model_open) return fail(c, BI_ERR_STATE, "bi_model_begin first");
    for (int64_t a = 0; a < c->A; ++a)
        if (!c->anchor_set[(size_t)a]) return fail(c, BI_ERR_STATE, "anchor %lld was never set", (long long)a);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->bb_source >= 0) {
        // N_c = sum_b n_model[c, i, b]; N(z) is linear in the corner weights (likelihood.py:645)
        hipLaunchKernelGGL(k_row_total, dim3((unsigned)c->A), dim3(kThreads), 0, c->stream, (const double*)c->nm.p, c->B,
                           c->Bp, (double*)c->nm_tot.p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(c->h_nm_tot.data(), c->nm_tot.p, (size_t)c->A * sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream));
    }
    {
        // row sums T_k and non-negativity of the templates: preconditions / constants of the sparse forms
        const int64_t rows = c->A * c->S;
        int rc = dev_alloc(c, c->scratch, (size_t)rows * 3 * sizeof(double));
        if (rc) return rc;
        hipLaunchKernelGGL(k_row_stats, dim3((unsigned)rows), dim3(kThreads), 0, c->stream, (const double*)c->ps.p, c->B,
                           c->Bp, (double*)c->scratch.p);
        HIP_TRY(c, hipGetLastError());
        std::vector<double> st((size_t)rows * 3);
        HIP_TRY(c, hipMemcpyAsync(st.data(), c->scratch.p, st.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->h_rowsum.assign((size_t)rows, 0.0);
        c->ps_nonneg = true;
        c->ps_finite = true;
        for (int64_t r = 0; r < rows; ++r) {
            c->h_rowsum[(size_t)r] = st[(size_t)r * 3];
            if (!(st[(size_t)r * 3 + 1] >= 0.0) || st[(size_t)r * 3 + 2] != 0.0) c->ps_nonneg = false;
            if (st[(size_t)r * 3 + 2] != 0.0) c->ps_finite = false;
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->model_open = false;
    c->model_ready = true;
    return BI_OK;
}

int bi_upload_model(bi_ctx* c, int d, const int32_t* n_anchor, const double* anchor_z, int S, int64_t B,
                    const double* ps, const double* mus, const double* n_model, int bb_source) {
    if (!c) return BI_ERR_INVALID;
    if (!ps || !mus) return fail(c, BI_ERR_INVALID, "ps / mus are NULL");
    if (bb_source >= 0 && !n_model) return fail(c, BI_ERR_INVALID, "bb_source given but n_model is NULL");
    int rc = bi_model_begin(c, d, n_anchor, anchor_z, S, B, bb_source);
    if (rc) return rc;
    // one strided copy for the whole tensor: rows are (anchor, source)
    if (B > 0) HIP_TRY(c, hipMemcpy2DAsync(c->ps.p, c->Bp * sizeof(double), ps, B * sizeof(double), B * sizeof(double),
                                (size_t)c->A * S, hipMemcpyHostToDevice, c->stream));
    std::copy(mus, mus + (size_t)c->A * S, c->h_mus.begin());
    if (bb_source >= 0) {
        // only row bb_source of every anchor is ever used (likelihood.py:643)
        if (B > 0) HIP_TRY(c, hipMemcpy2DAsync(c->nm.p, c->Bp * sizeof(double), n_model + (size_t)bb_source * B,
                                    (size_t)S * B * sizeof(double), B * sizeof(double), (size_t)c->A,
                                    hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::fill(c->anchor_set.begin(), c->anchor_set.end(), 1);
    return bi_model_end(c);
}

int bi_set_allow_negative(bi_ctx* c, const int32_t* allow) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!allow) return fail(c, BI_ERR_INVALID, "allow is NULL");
    c->allow_neg.assign(allow, allow + c->S);
    return BI_OK;
}

// ---- data ----------------------------------------------------------------------------------

int bi_upload_counts(bi_ctx* c, int64_t T, const double* counts) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (T < 1 || !counts) return fail(c, BI_ERR_INVALID, "need T >= 1 datasets and a counts pointer");
    if (c->unbinned) return fail(c, BI_ERR_STATE, "the context holds an unbinned likelihood: it has no binned counts");
    if (c->B < 1) return fail(c, BI_ERR_INVALID, "a binned likelihood needs at least one bin");
    HIP_TRY(c, hipSetDevice(c->device));
    c->data_ready = false;
    ++c->epoch;
    const size_t bytes = (size_t)T * c->Bp * sizeof(double);
    if ((rc = dev_alloc(c, c->counts, bytes))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->counts.p, 0, bytes, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(c->counts.p, c->Bp * sizeof(double), counts, c->B * sizeof(double),
                                c->B * sizeof(double), (size_t)T, hipMemcpyHostToDevice, c->stream));
    // sum_b lgamma(n+1) per dataset, on the device
    const int nblk = (int)std::min<int64_t>(256, (c->B + kThreads - 1) / kThreads);
    if ((rc = dev_alloc(c, c->lgsum, (size_t)T * sizeof(double)))) return rc;
    const int64_t chunk = 32768;  // datasets per launch (gridDim.y limit)
    if ((rc = dev_alloc(c, c->scratch, (size_t)std::min(T, chunk) * nblk * sizeof(double)))) return rc;
    for (int64_t t0 = 0; t0 < T; t0 += chunk) {
        const int64_t n = std::min(chunk, T - t0);
        hipLaunchKernelGGL(k_counts_lgamma, dim3(nblk, (unsigned)n), dim3(kThreads), 0, c->stream,
                           (const double*)c->counts.p + t0 * c->Bp, c->B, c->Bp, (double*)c->scratch.p, nblk);
        hipLaunchKernelGGL(k_rows_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)c->scratch.p, nblk, (double*)c->lgsum.p + t0, n);
    }
    HIP_TRY(c, hipGetLastError());
    c->h_lgsum.assign((size_t)T, 0.0);
    HIP_TRY(c, hipMemcpyAsync(c->h_lgsum.data(), c->lgsum.p, (size_t)T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->T = T;
    c->dense_counts = true;
    if ((rc = build_sparse_forms(c))) return rc;
    c->data_ready = true;
    return BI_OK;
}

// ---- planning ------------------------------------------------------------------------------

void bi_plan_destroy(bi_ctx* c, bi_plan* p) {
    if (!p) return;
    if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    free_plan_buffers(p);
    delete p;
}

int64_t bi_plan_bytes(const bi_plan* p) { return p ? p->bytes : 0; }
int64_t bi_plan_launches(const bi_plan* p) { return p ? p->launches : 0; }

int bi_plan_points(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset,
                   bi_plan** out) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!out) return fail(c, BI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (P < 0) return fail(c, BI_ERR_INVALID, "P < 0");
    if (c->d > 0 && P > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    HIP_TRY(c, hipSetDevice(c->device));

    const int S = c->S, d = c->d;
    const bool bb = c->bb_source >= 0;
    const int nc = 1 << (int)c->eff_axes.size();
    const int n0 = bb ? nc * (S - 1) : nc * S;
    const int n1 = bb ? nc : 0, n2 = bb ? nc : 0;
    const int NS = n0 + n1 + n2;

    bool any_neg = false;
    for (int q = 0; q < S; ++q) any_neg |= (c->allow_neg[(size_t)q] != 0);
    const bool sparse = c->sparse && c->compact_ready && !bb && !any_neg && !c->unbinned;
    const int64_t n_rows = c->A * S;
    if (!sparse && !c->dense_counts)
        return fail(c, BI_ERR_STATE, "the datasets exist only as non-empty-bin lists (device-generated toys): point "
                                     "evaluations need the compacted templates (sparse mode, budget) or bi_eval_datasets");

    bi_plan* plan = new bi_plan();
    plan->P = P;
    plan->sparse = sparse;
    plan->epoch = c->epoch;
    plan->h_status.assign((size_t)P, 0);

    struct Pt { int64_t key; int64_t idx; };
    std::vector<Pt> pts;
    pts.reserve((size_t)P);
    std::vector<int64_t> bad;
    std::vector<PointGeom> geom((size_t)P);
    std::vector<double> rates((size_t)P * S);
    std::vector<double> ones((size_t)S, 1.0);

    for (int64_t p = 0; p < P; ++p) {
        const int64_t ds = dataset ? dataset[p] : 0;
        if (ds < 0 || ds >= c->T) { plan->h_status[p] |= BI_ST_BAD_DATASET; bad.push_back(p); continue; }
        PointGeom& g = geom[(size_t)p];
        if (!point_geometry(c, z ? z + p * d : nullptr, g)) { plan->h_status[p] |= BI_ST_OUT_OF_BOUNDS; bad.push_back(p); continue; }
        double* r = &rates[(size_t)p * S];
        interp_mus(c, g, r);
        const double* rs = rate_scale ? rate_scale + p * S : ones.data();
        for (int s = 0; s < S; ++s) r[s] *= rs[s];
        if (!rates_physical(c, r)) { plan->h_status[p] |= BI_ST_UNPHYSICAL; bad.push_back(p); continue; }
        pts.push_back({g.cell_anchor * c->T + ds, p});
    }
    std::stable_sort(pts.begin(), pts.end(), [](const Pt& a, const Pt& b) { return a.key < b.key; });

    // chop every (cell, dataset) group into items of the available G classes
    const int classG[5] = {1, 2, 4, 8, 16};
    struct HostClass { std::vector<int64_t> rowoff, cnt_off; std::vector<double> coef, aux, slot_lg; std::vector<int32_t> tiles; std::vector<int64_t> perm; int64_t bytes = 0; };
    HostClass hc[5];
    const int maxg = bb ? (int)std::min<int64_t>(c->max_group, 8) : (int)c->max_group;  // G=16 with BB spills past 256 VGPRs
    size_t i = 0;
    std::vector<int64_t> corner_off((size_t)nc);
    for (int k = 0; k < nc; ++k) corner_off[(size_t)k] = corner_offset(c, k);
    std::vector<char> anchor_used((size_t)c->A, 0);
    bool reuse = false;
    while (i < pts.size()) {
        size_t j = i;
        while (j < pts.size() && pts[j].key == pts[i].key) ++j;
        size_t n = j - i;
        while (n > 0) {
            const int G = pick_class((int)std::min<size_t>(n, (size_t)maxg), maxg);
            const int take = (int)std::min<size_t>(n, (size_t)G);
            int ci = 0;
            while (classG[ci] != G) ++ci;
            HostClass& h = hc[ci];
            const int64_t p0 = pts[i].idx;
            const int64_t cell = geom[(size_t)p0].cell_anchor;
            const int64_t ds = pts[i].key % c->T;
            for (int corner = 0; corner < nc; ++corner) {
                char& u = anchor_used[(size_t)(cell + corner_off[(size_t)corner])];
                if (u) reuse = true;
                u = 1;
            }
            const int64_t row_stride = sparse ? c->h_c_np[(size_t)ds] : c->Bp;
            const int64_t row_base = sparse ? c->h_c_off[(size_t)ds] : 0;
            const size_t ro = h.rowoff.size();
            h.rowoff.resize(ro + NS);
            // stream rows: [n0] (corner, source != bb) ; [n1] (corner, bb source) ; [n2] n_model corner rows
            int k0 = 0;
            for (int corner = 0; corner < nc; ++corner)
                for (int s = 0; s < S; ++s) {
                    if (bb && s == c->bb_source) continue;
                    h.rowoff[ro + k0++] = row_base + ((cell + corner_off[(size_t)corner]) * S + s) * row_stride;
                }
            for (int corner = 0; corner < n1; ++corner)
                h.rowoff[ro + n0 + corner] = ((cell + corner_off[(size_t)corner]) * S + c->bb_source) * c->Bp;
            for (int corner = 0; corner < n2; ++corner)
                h.rowoff[ro + n0 + n1 + corner] = (cell + corner_off[(size_t)corner]) * c->Bp;
            const size_t co = h.coef.size();
            h.coef.resize(co + (size_t)NS * G, 0.0);
            const size_t ao = h.aux.size();
            h.aux.resize(ao + (size_t)G * 2, 1.0);
            const size_t po = h.perm.size();
            h.perm.resize(po + G, -1);
            h.cnt_off.push_back(sparse ? c->h_cnt_off[(size_t)ds] : ds * c->Bp);
            h.tiles.push_back((int32_t)(row_stride / kTile));
            h.bytes += (int64_t)sizeof(double) * ((int64_t)NS + 1) * (sparse ? row_stride : c->B);
            h.slot_lg.resize(po + G, c->h_lgsum[(size_t)ds]);
            for (int g = 0; g < take; ++g) {
                const int64_t p = pts[i + g].idx;
                const PointGeom& pg = geom[(size_t)p];
                const double* r = &rates[(size_t)p * S];
                int k = 0;
                for (int corner = 0; corner < nc; ++corner)
                    for (int s = 0; s < S; ++s) {
                        if (bb && s == c->bb_source) continue;
                        h.coef[co + (size_t)(k++) * G + g] = pg.w[(size_t)corner] * r[s];
                    }
                for (int corner = 0; corner < n1; ++corner) h.coef[co + (size_t)(n0 + corner) * G + g] = pg.w[(size_t)corner];
                for (int corner = 0; corner < n2; ++corner) h.coef[co + (size_t)(n0 + n1 + corner) * G + g] = pg.w[(size_t)corner];
                if (bb) {
                    double Ntot = 0.0;
                    for (int corner = 0; corner < nc; ++corner) {
                        const double term = c->h_nm_tot[(size_t)(pg.cell_anchor + corner_off[(size_t)corner])] * pg.w[(size_t)corner];
                        Ntot = Ntot + term;
                    }
                    h.aux[ao + (size_t)g * 2 + 0] = r[c->bb_source] / Ntot;  // p_calibration, likelihood.py:645
                    h.aux[ao + (size_t)g * 2 + 1] = Ntot;
                }
                h.perm[po + g] = p;
                if (c->unbinned) {   // ll = -sum_s mu_s + sum_e log(...)   (likelihood.py:690)
                    double rsum = 0.0;
                    for (int s = 0; s < S; ++s) rsum += r[s];
                    h.slot_lg[po + g] = rsum;
                }
                if (sparse) {
                    // minus sum_k coef_k * (sum of row k over the empty bins of this dataset)
                    double zsum = 0.0;
                    int kk = 0;
                    for (int corner = 0; corner < nc; ++corner)
                        for (int s = 0; s < S; ++s) {
                            const int64_t row = (cell + corner_off[(size_t)corner]) * S + s;
                            zsum += h.coef[co + (size_t)(kk++) * G + g] * c->h_Tz[(size_t)(ds * n_rows + row)];
                        }
                    h.slot_lg[po + g] += zsum;
                }
            }
            i += take;
            n -= take;
        }
    }

    // grid shape: enough blocks to fill the chip, few enough that partial buffers stay small
    const int n_tiles = n_tiles_of(c);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    int64_t total_items = 0;
    for (auto& h : hc) total_items += (int64_t)h.tiles.size();

    for (int ci = 0; ci < 5; ++ci) {
        HostClass& h = hc[ci];
        if (h.tiles.empty()) continue;
        bi_plan::Class k;
        k.G = classG[ci];
        k.n_items = (int64_t)h.tiles.size();
        const int64_t max_tiles = sparse ? *std::max_element(h.tiles.begin(), h.tiles.end()) : n_tiles;
        int64_t nbx = std::min<int64_t>(max_tiles, std::max<int64_t>(1, (4 * slots + total_items - 1) / total_items));
        if (total_items == 1) nbx = std::min<int64_t>(max_tiles, slots);
        k.nbx = (int)nbx;
        if ((rc = dev_upload(c, k.rowoff, h.rowoff)) || (rc = dev_upload(c, k.coef, h.coef)) ||
            (rc = dev_upload(c, k.aux, h.aux)) || (rc = dev_upload(c, k.item_cnt, h.cnt_off)) ||
            (rc = dev_upload(c, k.item_tiles, h.tiles)) ||
            (rc = dev_upload(c, k.perm, h.perm)) || (rc = dev_upload(c, k.slot_lg, h.slot_lg)) ||
            (rc = dev_alloc(c, k.partial, (size_t)k.n_items * k.nbx * k.G * sizeof(double))) ||
            (rc = dev_alloc(c, k.pflags, (size_t)k.n_items * k.nbx * k.G * sizeof(unsigned)))) {
            plan->classes.push_back(k);
            free_plan_buffers(plan);
            delete plan;
            return rc;
        }
        plan->bytes += h.bytes;
        plan->launches += (k.n_items + 65534) / 65535;
        plan->classes.push_back(k);
    }
    plan->no_reuse = !reuse;
    plan->n_bad = (int64_t)bad.size();
    if ((rc = dev_upload(c, plan->bad_idx, bad)) || (rc = dev_alloc(c, plan->out, (size_t)std::max<int64_t>(P, 1) * sizeof(double))) ||
        (rc = dev_upload(c, plan->status, plan->h_status))) {
        free_plan_buffers(plan);
        delete plan;
        return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // host staging vectors die with this scope
    *out = plan;
    return BI_OK;
}

int bi_run_plan(bi_ctx* c, bi_plan* plan, double* out_dev) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!plan) return fail(c, BI_ERR_INVALID, "plan is NULL");
    if (plan->epoch != c->epoch) return fail(c, BI_ERR_STATE, "plan is stale: model or data were uploaded after it was made");
    HIP_TRY(c, hipSetDevice(c->device));
    double* out = out_dev ? out_dev : (double*)plan->out.p;
    const bool bb = c->bb_source >= 0;
    const int nc = 1 << (int)c->eff_axes.size();
    LaunchArgs a{};
    a.ps = plan->sparse ? (const double*)c->ps_c.p : (const double*)c->ps.p;
    a.nm = (const double*)c->nm.p;
    a.counts = plan->sparse ? (const double*)c->cnt_c.p : (const double*)c->counts.p;
    a.B = c->B; a.Bp = c->Bp;
    a.outlier = c->outlier;
    a.n0 = bb ? nc * (c->S - 1) : nc * c->S;
    a.n1 = bb ? nc : 0; a.n2 = bb ? nc : 0;
    a.n_tiles = n_tiles_of(c);
    const int NS = a.n0 + a.n1 + a.n2;
    for (auto& k : plan->classes) {
        for (int64_t i0 = 0; i0 < k.n_items; i0 += 65535) {
            const int64_t ni = std::min<int64_t>(65535, k.n_items - i0);
            LaunchArgs b = a;
            b.rowoff = (const int64_t*)k.rowoff.p + i0 * NS;
            b.coef = (const double*)k.coef.p + i0 * NS * k.G;
            b.aux = (const double*)k.aux.p + i0 * k.G * 2;
            b.item_cnt = (const int64_t*)k.item_cnt.p + i0;
            b.item_tiles = (const int32_t*)k.item_tiles.p + i0;
            b.partial = (double*)k.partial.p + i0 * k.nbx * k.G;
            b.pflags = (unsigned*)k.pflags.p + i0 * k.nbx * k.G;
            const bool nt = !plan->sparse && (c->nt_loads == 1 || (c->nt_loads == 2 && plan->no_reuse));
            launch_morph_g(c, k.G, b, dim3((unsigned)k.nbx, (unsigned)ni), bb, nt);
            const int64_t n_slots = ni * k.G;
            const int lanes = k.nbx > 64 ? kThreads : 64;
            const int per_block = kThreads / lanes;
            hipLaunchKernelGGL(k_finish, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(kThreads), 0,
                               c->stream, (const double*)b.partial, (const unsigned*)b.pflags, k.nbx, k.G, lanes, n_slots,
                               (const int64_t*)k.perm.p + i0 * k.G, (const double*)k.slot_lg.p + i0 * k.G, out,
                               (int32_t*)plan->status.p);
        }
    }
    if (plan->n_bad > 0)
        hipLaunchKernelGGL(k_fill_const, dim3((unsigned)((plan->n_bad + 255) / 256)), dim3(256), 0, c->stream, out,
                           (const int64_t*)plan->bad_idx.p, plan->n_bad, -std::numeric_limits<double>::infinity());
    HIP_TRY(c, hipGetLastError());
    return BI_OK;
}

int bi_plan_read(bi_ctx* c, bi_plan* plan, double* out, int32_t* status) {
    if (!c || !plan) return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (out && plan->P)
        HIP_TRY(c, hipMemcpyAsync(out, plan->out.p, (size_t)plan->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status && plan->P)
        HIP_TRY(c, hipMemcpyAsync(status, plan->status.p, (size_t)plan->P * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BI_OK;
}

int bi_eval(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, double* out,
            int32_t* status) {
    if (!c) return BI_ERR_INVALID;
    if (P > 0 && !out) return fail(c, BI_ERR_INVALID, "out is NULL");
    if (P == 1) {
        int rc1 = check_ready(c, true);
        if (rc1) return rc1;
        if (c->d > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
        HIP_TRY(c, hipSetDevice(c->device));
        if (status) *status = 0;
        return eval_single(c, z, rate_scale, dataset ? dataset[0] : 0, out, status);
    }
    bi_plan* plan = nullptr;
    int rc = bi_plan_points(c, P, z, rate_scale, dataset, &plan);
    if (rc) return rc;
    rc = bi_run_plan(c, plan, nullptr);
    if (!rc) rc = bi_plan_read(c, plan, out, status);
    bi_plan_destroy(c, plan);
    return rc;
}


// ---- value + analytic gradient in one pass ----------------------------------------------------

int bi_eval_grad(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, double* ll,
                 double* grad, int32_t* status) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (c->bb_source >= 0) return fail(c, BI_ERR_INVALID, "bi_eval_grad is not available with Beeston-Barlow");
    if (c->unbinned) return fail(c, BI_ERR_INVALID, "bi_eval_grad is implemented for binned likelihoods only");
    if (P < 0 || (P > 0 && (!ll || !grad))) return fail(c, BI_ERR_INVALID, "bad P / output pointers");
    if (c->d > 0 && P > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    const int S = c->S, d = c->d;
    const int W = 1 + d + S;
    if (W > kMaxG) return fail(c, BI_ERR_INVALID, "1 + d + S = %d exceeds %d gradient columns", W, kMaxG);
    HIP_TRY(c, hipSetDevice(c->device));
    const int G = std::max(2, pick_class(W, kMaxG));
    const int de = (int)c->eff_axes.size();
    const int nc = 1 << de, NS = nc * S;
    bool any_neg = false;
    for (int q = 0; q < S; ++q) any_neg |= (c->allow_neg[(size_t)q] != 0);
    const bool sparse = c->sparse && c->compact_ready && !any_neg;
    if (!sparse && !c->dense_counts) return fail(c, BI_ERR_STATE, "dataset counts are not resident in dense form");
    const int64_t n_rows = c->A * S;
    const double ninf = -std::numeric_limits<double>::infinity();
    const double qnan = std::numeric_limits<double>::quiet_NaN();

    std::vector<int64_t> rowoff, cnt_off, perm;
    std::vector<double> coef, slot_lg;
    std::vector<int32_t> tiles;
    std::vector<int64_t> live;  // point index of every item
    std::vector<double> ones((size_t)S, 1.0), mus((size_t)S), dmus((size_t)S * std::max(d, 1));
    std::vector<double> dw((size_t)nc * std::max(de, 1));
    std::vector<int64_t> corner_off((size_t)nc);
    for (int k = 0; k < nc; ++k) corner_off[(size_t)k] = corner_offset(c, k);
    int max_tiles = 1;
    int64_t bytes = 0;
    for (int64_t p = 0; p < P; ++p) {
        if (status) status[p] = 0;
        ll[p] = ninf;
        for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = qnan;
        const int64_t ds = dataset ? dataset[p] : 0;
        if (ds < 0 || ds >= c->T) { if (status) status[p] = BI_ST_BAD_DATASET; continue; }
        PointGeom g;
        if (!point_geometry(c, z ? z + p * d : nullptr, g)) { if (status) status[p] = BI_ST_OUT_OF_BOUNDS; continue; }
        interp_mus(c, g, mus.data());
        const double* rs = rate_scale ? rate_scale + p * S : ones.data();
        std::vector<double> r((size_t)S);
        for (int s = 0; s < S; ++s) r[(size_t)s] = mus[(size_t)s] * rs[s];
        if (!rates_physical(c, r.data())) { if (status) status[p] = BI_ST_UNPHYSICAL; continue; }
        // d w_c / d z_i for the effective axes: (+-1/delta_i) * prod_{j != i} w^(j)
        for (int corner = 0; corner < nc; ++corner)
            for (int i = 0; i < de; ++i) {
                const int ax = c->eff_axes[(size_t)i];
                double v = (((corner >> (de - 1 - i)) & 1) ? 1.0 : -1.0) * g.inv_delta[ax];
                for (int j = 0; j < de; ++j) {
                    if (j == i) continue;
                    const double t = g.t[c->eff_axes[(size_t)j]];
                    v *= ((corner >> (de - 1 - j)) & 1) ? t : (1 - t);
                }
                dw[(size_t)corner * de + i] = v;
            }
        // d mus_s / d z_i
        for (int i = 0; i < de; ++i)
            for (int s = 0; s < S; ++s) {
                double v = 0.0;
                for (int corner = 0; corner < nc; ++corner)
                    v += dw[(size_t)corner * de + i] * c->h_mus[(size_t)((g.cell_anchor + corner_off[(size_t)corner]) * S + s)];
                dmus[(size_t)i * S + s] = v;
            }
        const int64_t row_stride = sparse ? c->h_c_np[(size_t)ds] : c->Bp;
        const int64_t row_base = sparse ? c->h_c_off[(size_t)ds] : 0;
        const size_t ro = rowoff.size(), co = coef.size(), po = perm.size();
        rowoff.resize(ro + NS);
        coef.resize(co + (size_t)NS * G, 0.0);
        perm.resize(po + G, -1);
        slot_lg.resize(po + G, 0.0);
        int k = 0;
        for (int corner = 0; corner < nc; ++corner)
            for (int s = 0; s < S; ++s, ++k) {
                const int64_t row = (g.cell_anchor + corner_off[(size_t)corner]) * S + s;
                rowoff[ro + k] = row_base + row * row_stride;
                double* col = &coef[co + (size_t)k * G];
                const double w = g.w[(size_t)corner];
                col[0] = w * r[(size_t)s];
                for (int i = 0; i < de; ++i)   // total derivative w.r.t. z: through the weights and through mus(z)
                    col[1 + c->eff_axes[(size_t)i]] = dw[(size_t)corner * de + i] * r[(size_t)s] + w * dmus[(size_t)i * S + s] * rs[s];
                col[1 + d + s] = w * mus[(size_t)s];
                if (sparse) {
                    const double tz = c->h_Tz[(size_t)(ds * n_rows + row)];
                    for (int q = 0; q < W; ++q) slot_lg[po + q] += col[q] * tz;
                }
            }
        slot_lg[po] += c->h_lgsum[(size_t)ds];
        for (int q = 0; q < W; ++q) perm[po + q] = (int64_t)live.size() * W + q;
        cnt_off.push_back(sparse ? c->h_cnt_off[(size_t)ds] : ds * c->Bp);
        tiles.push_back((int32_t)(row_stride / kTile));
        max_tiles = std::max(max_tiles, tiles.back());
        bytes += (int64_t)sizeof(double) * ((int64_t)NS + 1) * (sparse ? row_stride : c->B);
        live.push_back(p);
    }
    const int64_t n_items = (int64_t)live.size();
    if (n_items == 0) return BI_OK;
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nbx = (int)std::min<int64_t>(max_tiles, n_items == 1 ? slots : std::max<int64_t>(1, (4 * slots + n_items - 1) / n_items));
    DevBuf d_row, d_coef, d_cnt, d_tiles, d_perm, d_lg, d_part, d_flag, d_out;
    auto cleanup = [&]() { dev_free(d_row); dev_free(d_coef); dev_free(d_cnt); dev_free(d_tiles); dev_free(d_perm);
                           dev_free(d_lg); dev_free(d_part); dev_free(d_flag); dev_free(d_out); };
    if ((rc = dev_upload(c, d_row, rowoff)) || (rc = dev_upload(c, d_coef, coef)) || (rc = dev_upload(c, d_cnt, cnt_off)) ||
        (rc = dev_upload(c, d_tiles, tiles)) || (rc = dev_upload(c, d_perm, perm)) || (rc = dev_upload(c, d_lg, slot_lg)) ||
        (rc = dev_alloc(c, d_part, (size_t)n_items * nbx * G * sizeof(double))) ||
        (rc = dev_alloc(c, d_flag, (size_t)n_items * nbx * G * sizeof(unsigned))) ||
        (rc = dev_alloc(c, d_out, (size_t)n_items * W * sizeof(double)))) {
        cleanup();
        return rc;
    }
    LaunchArgs a{};
    a.ps = sparse ? (const double*)c->ps_c.p : (const double*)c->ps.p;
    a.counts = sparse ? (const double*)c->cnt_c.p : (const double*)c->counts.p;
    a.B = c->B; a.Bp = c->Bp; a.n0 = NS; a.n_tiles = max_tiles;
    const bool nt = !sparse && (c->nt_loads == 1 || (c->nt_loads == 2 && n_items == 1));
    for (int64_t i0 = 0; i0 < n_items; i0 += 65535) {
        const int64_t ni = std::min<int64_t>(65535, n_items - i0);
        LaunchArgs b = a;
        b.rowoff = (const int64_t*)d_row.p + i0 * NS;
        b.coef = (const double*)d_coef.p + i0 * NS * G;
        b.item_cnt = (const int64_t*)d_cnt.p + i0;
        b.item_tiles = (const int32_t*)d_tiles.p + i0;
        b.partial = (double*)d_part.p + i0 * nbx * G;
        b.pflags = (unsigned*)d_flag.p + i0 * nbx * G;
        launch_morph_grad(c, G, b, dim3((unsigned)nbx, (unsigned)ni), nt);
        const int64_t n_slots = ni * G;
        const int lanes = nbx > 64 ? kThreads : 64;
        const int per_block = kThreads / lanes;
        hipLaunchKernelGGL(k_finish, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(kThreads), 0, c->stream,
                           (const double*)b.partial, (const unsigned*)b.pflags, nbx, G, lanes, n_slots,
                           (const int64_t*)d_perm.p + i0 * G, (const double*)d_lg.p + i0 * G, (double*)d_out.p,
                           (int32_t*)nullptr);
    }
    std::vector<double> h_out((size_t)n_items * W);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_out.data(), d_out.p, h_out.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_grad: %s", hipGetErrorString(e));
    for (int64_t i = 0; i < n_items; ++i) {
        const int64_t p = live[(size_t)i];
        ll[p] = h_out[(size_t)i * W];
        for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = h_out[(size_t)i * W + 1 + j];
    }
    (void)bytes;
    return BI_OK;
}

// ---- toy-MC form ---------------------------------------------------------------------------

int bi_eval_datasets(bi_ctx* c, const double* z, const double* rate_scale, int64_t t0, int64_t t1, double* out,
                     int32_t* status) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (c->bb_source >= 0) return fail(c, BI_ERR_INVALID, "bi_eval_datasets is not available with Beeston-Barlow");
    if (c->unbinned) return fail(c, BI_ERR_INVALID, "bi_eval_datasets needs a binned likelihood");
    if (t0 < 0 || t1 > c->T || t0 > t1) return fail(c, BI_ERR_INVALID, "dataset range [%lld,%lld) outside [0,%lld)", (long long)t0, (long long)t1, (long long)c->T);
    if (c->d > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    if (t1 > t0 && !out) return fail(c, BI_ERR_INVALID, "out is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t n = t1 - t0;
    if (status) *status = 0;
    const double ninf = -std::numeric_limits<double>::infinity();
    PointGeom g;
    if (!point_geometry(c, z, g)) {
        if (status) *status = BI_ST_OUT_OF_BOUNDS;
        std::fill(out, out + n, ninf);
        return BI_OK;
    }
    std::vector<double> r((size_t)c->S);
    interp_mus(c, g, r.data());
    if (rate_scale) for (int s = 0; s < c->S; ++s) r[(size_t)s] *= rate_scale[s];
    if (!rates_physical(c, r.data())) {
        if (status) *status = BI_ST_UNPHYSICAL;
        std::fill(out, out + n, ninf);
        return BI_OK;
    }
    const int nc = (int)g.w.size();
    const int NS = nc * c->S;
    std::vector<int64_t> rowoff((size_t)NS);
    std::vector<double> coef((size_t)NS);
    int k = 0;
    for (int corner = 0; corner < nc; ++corner)
        for (int s = 0; s < c->S; ++s) {
            rowoff[(size_t)k] = ((g.cell_anchor + corner_offset(c, corner)) * c->S + s) * c->Bp;
            coef[(size_t)k++] = g.w[(size_t)corner] * r[(size_t)s];
        }
    const int n_tiles = n_tiles_of(c);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nmu = (int)std::min<int64_t>(n_tiles, slots);
    DevBuf d_row, d_coef, d_out;
    auto cleanup = [&]() { dev_free(d_row); dev_free(d_coef); dev_free(d_out); };
    const bool csr = (c->sparse && c->csr_ready) || !c->dense_counts;
    if (csr && !c->csr_ready) return fail(c, BI_ERR_STATE, "no counts resident");
    const int64_t chunk = csr ? 1048576 : 16384;
    const int nbx = csr ? 1 : (int)std::min<int64_t>(n_tiles, std::max<int64_t>(1, 4 * slots / std::max<int64_t>(1, std::min(n, chunk))));
    if ((rc = dev_upload(c, d_row, rowoff)) || (rc = dev_upload(c, d_coef, coef)) ||
        (rc = dev_alloc(c, c->logmu, (size_t)c->Bp * sizeof(double))) ||
        (rc = dev_alloc(c, c->scratch, (size_t)nmu * sizeof(double) + (size_t)nmu * sizeof(unsigned) + 64)) ||
        (rc = dev_alloc(c, c->scratch2, (size_t)std::min(n, chunk) * nbx * sizeof(double))) ||
        (rc = dev_alloc(c, d_out, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) {
        cleanup();
        return rc;
    }
    LaunchArgs a{};
    a.ps = (const double*)c->ps.p;
    a.rowoff = (const int64_t*)d_row.p;
    a.coef = (const double*)d_coef.p;
    a.partial = (double*)c->scratch.p;
    a.pflags = (unsigned*)((char*)c->scratch.p + (((size_t)nmu * sizeof(double) + 63) / 64) * 64);
    a.B = c->B; a.Bp = c->Bp; a.n0 = NS; a.n_tiles = n_tiles;
    {
        EventScope ev(c);
        hipLaunchKernelGGL(k_morph_logmu, dim3((unsigned)nmu), dim3(kThreads), 0, c->stream, a, (double*)c->logmu.p, 0);
    }
    for (int64_t s0 = 0; s0 < n; s0 += chunk) {
        const int64_t ni = std::min(chunk, n - s0);
        {
            EventScope ev(c);
            if (csr)
                hipLaunchKernelGGL(k_dataset_dot_csr, dim3((unsigned)ni), dim3(kThreads), 0, c->stream,
                                   (const int32_t*)c->nz_idx.p, (const double*)c->nz_n.p, (const int64_t*)c->nz_off.p,
                                   (const double*)c->logmu.p, t0 + s0, (double*)c->scratch2.p);
            else
                hipLaunchKernelGGL(k_dataset_dot, dim3((unsigned)nbx, (unsigned)ni), dim3(kThreads), 0, c->stream,
                                   (const double*)c->counts.p, (const double*)c->logmu.p, c->Bp, n_tiles, t0 + s0,
                                   (double*)c->scratch2.p);
        }
        hipLaunchKernelGGL(k_dataset_finish, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)c->scratch2.p, nbx, (const double*)a.partial, (const unsigned*)a.pflags, nmu,
                           (const double*)c->lgsum.p, t0 + s0, ni, (double*)d_out.p + s0);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && n) e = hipMemcpyAsync(out, d_out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_datasets: %s", hipGetErrorString(e));
    return BI_OK;
}


// ---- toy-MC generation -------------------------------------------------------------------------

int bi_generate_toys(bi_ctx* c, const double* z, const double* rate_scale, int64_t T, uint64_t seed) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (T < 1) return fail(c, BI_ERR_INVALID, "need T >= 1 toys");
    if (c->B < 1) return fail(c, BI_ERR_INVALID, "a binned likelihood needs at least one bin");
    if (c->d > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    PointGeom g;
    if (!point_geometry(c, z, g)) return fail(c, BI_ERR_INVALID, "toy generation point is outside the anchor box");
    std::vector<double> r((size_t)c->S);
    interp_mus(c, g, r.data());
    if (rate_scale) for (int s = 0; s < c->S; ++s) r[(size_t)s] *= rate_scale[s];
    for (int s = 0; s < c->S; ++s)
        if (!(r[(size_t)s] >= 0.0 && r[(size_t)s] < std::numeric_limits<double>::infinity()))
            return fail(c, BI_ERR_INVALID, "toy generation needs rates in [0, inf)");
    c->data_ready = false;
    c->dense_counts = false;
    c->csr_ready = c->compact_ready = false;
    ++c->epoch;
    dev_free(c->counts);  // the toys exist as non-empty-bin lists only
    const int nc = (int)g.w.size(), NS = nc * c->S;
    std::vector<int64_t> rowoff((size_t)NS);
    std::vector<double> coef((size_t)NS);
    int k = 0;
    for (int corner = 0; corner < nc; ++corner)
        for (int s = 0; s < c->S; ++s) {
            rowoff[(size_t)k] = ((g.cell_anchor + corner_offset(c, corner)) * c->S + s) * c->Bp;
            coef[(size_t)k++] = g.w[(size_t)corner] * r[(size_t)s];
        }
    const int n_tiles = n_tiles_of(c);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nmu = (int)std::min<int64_t>(n_tiles, slots);
    const int64_t B = c->B;
    const int nchunks = (int)((B + kNzChunk - 1) / kNzChunk);
    DevBuf d_row, d_coef, d_cnt, d_off, d_lgp;
    auto cleanup = [&]() { dev_free(d_row); dev_free(d_coef); dev_free(d_cnt); dev_free(d_off); dev_free(d_lgp); };
    if ((rc = dev_upload(c, d_row, rowoff)) || (rc = dev_upload(c, d_coef, coef)) ||
        (rc = dev_alloc(c, c->logmu, (size_t)c->Bp * sizeof(double))) ||
        (rc = dev_alloc(c, c->scratch, (size_t)nmu * sizeof(double) + (size_t)nmu * sizeof(unsigned) + 64)) ||
        (rc = dev_alloc(c, d_cnt, (size_t)T * nchunks * sizeof(int32_t))) ||
        (rc = dev_alloc(c, d_lgp, (size_t)T * nchunks * sizeof(double))) || (rc = dev_alloc(c, c->lgsum, (size_t)T * sizeof(double)))) {
        cleanup();
        return rc;
    }
    LaunchArgs a{};
    a.ps = (const double*)c->ps.p;
    a.rowoff = (const int64_t*)d_row.p;
    a.coef = (const double*)d_coef.p;
    a.partial = (double*)c->scratch.p;
    a.pflags = (unsigned*)((char*)c->scratch.p + (((size_t)nmu * sizeof(double) + 63) / 64) * 64);
    a.B = c->B; a.Bp = c->Bp; a.n0 = NS; a.n_tiles = n_tiles;
    hipLaunchKernelGGL(k_morph_logmu, dim3((unsigned)nmu), dim3(kThreads), 0, c->stream, a, (double*)c->logmu.p, 1);
    const double* mu = (const double*)c->logmu.p;
    const int64_t tchunk = 32768;
    for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
        const int64_t n = std::min(tchunk, T - t0);
        hipLaunchKernelGGL(k_toy_count, dim3((unsigned)nchunks, (unsigned)n), dim3(kThreads), 0, c->stream, mu, B, seed, t0,
                           (int32_t*)d_cnt.p + t0 * nchunks, nchunks);
    }
    std::vector<int32_t> h_cnt((size_t)T * nchunks);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt.data(), d_cnt.p, h_cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return fail(c, BI_ERR_HIP, "toy count: %s", hipGetErrorString(e)); }
    std::vector<int64_t> h_off(h_cnt.size());
    c->h_nz_off.assign((size_t)T + 1, 0);
    int64_t run = 0;
    for (int64_t t = 0; t < T; ++t) {
        c->h_nz_off[(size_t)t] = run;
        for (int q = 0; q < nchunks; ++q) { h_off[(size_t)t * nchunks + q] = run; run += h_cnt[(size_t)t * nchunks + q]; }
    }
    c->h_nz_off[(size_t)T] = run;
    if ((rc = dev_upload(c, d_off, h_off)) || (rc = dev_alloc(c, c->nz_idx, (size_t)std::max<int64_t>(run, 1) * sizeof(int32_t))) ||
        (rc = dev_alloc(c, c->nz_n, (size_t)std::max<int64_t>(run, 1) * sizeof(double))) || (rc = dev_upload(c, c->nz_off, c->h_nz_off))) {
        cleanup();
        return rc;
    }
    for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
        const int64_t n = std::min(tchunk, T - t0);
        hipLaunchKernelGGL(k_toy_scatter, dim3((unsigned)nchunks, (unsigned)n), dim3(kThreads), 0, c->stream, mu, B, seed, t0,
                           (const int64_t*)d_off.p + t0 * nchunks, nchunks, (int32_t*)c->nz_idx.p, (double*)c->nz_n.p,
                           (double*)d_lgp.p + t0 * nchunks);
        hipLaunchKernelGGL(k_rows_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)d_lgp.p + t0 * nchunks, nchunks, (double*)c->lgsum.p + t0, n);
    }
    c->h_lgsum.assign((size_t)T, 0.0);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_lgsum.data(), c->lgsum.p, (size_t)T * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "toy scatter: %s", hipGetErrorString(e));
    c->T = T;
    c->csr_ready = true;
    if ((rc = build_compact_templates(c))) return rc;   // per-toy point evaluations, when the budget allows
    c->data_ready = true;
    return BI_OK;
}

int bi_download_counts(bi_ctx* c, int64_t t, double* out) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (t < 0 || t >= c->T || !out) return fail(c, BI_ERR_INVALID, "dataset %lld outside [0,%lld) or out is NULL", (long long)t, (long long)c->T);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->dense_counts) {
        HIP_TRY(c, hipMemcpyAsync(out, (const double*)c->counts.p + t * c->Bp, (size_t)c->B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return BI_OK;
    }
    if (!c->csr_ready) return fail(c, BI_ERR_STATE, "no counts resident");
    if ((rc = dev_alloc(c, c->logmu, (size_t)c->Bp * sizeof(double)))) return rc;
    const int64_t lo = c->h_nz_off[(size_t)t], nnz = c->h_nz_off[(size_t)t + 1] - lo;
    HIP_TRY(c, hipMemsetAsync(c->logmu.p, 0, (size_t)c->B * sizeof(double), c->stream));
    if (nnz > 0)
        hipLaunchKernelGGL(k_csr_to_dense, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, c->stream,
                           (const int32_t*)c->nz_idx.p + lo, (const double*)c->nz_n.p + lo, nnz, (double*)c->logmu.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->logmu.p, (size_t)c->B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BI_OK;
}


// ---- extended unbinned likelihood -----------------------------------------------------------------

int bi_set_unbinned(bi_ctx* c, double outlier_likelihood) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (c->bb_source >= 0) return fail(c, BI_ERR_INVALID, "Beeston-Barlow applies to binned likelihoods only");
    if (!c->ps_finite)
        return fail(c, BI_ERR_INVALID, "pdf values at the events must be finite (the reference's nansum over sources, "
                                       "likelihood.py:686, is not reproduced)");
    HIP_TRY(c, hipSetDevice(c->device));
    ++c->epoch;
    c->unbinned = true;
    c->outlier = outlier_likelihood;
    c->csr_ready = c->compact_ready = false;
    // one pseudo dataset with zero lgamma sum; the counts row is never read in this mode
    if ((rc = dev_alloc(c, c->counts, (size_t)c->Bp * sizeof(double)))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->counts.p, 0, (size_t)c->Bp * sizeof(double), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->T = 1;
    c->h_lgsum.assign(1, 0.0);
    c->dense_counts = true;
    c->data_ready = true;
    return BI_OK;
}

// ---- compatibility mode --------------------------------------------------------------------

int bi_interpolate(bi_ctx* c, int which, const double* z, double* out) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!out) return fail(c, BI_ERR_INVALID, "out is NULL");
    if (c->d > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    if (which < 0 || which > 2) return fail(c, BI_ERR_INVALID, "which must be 0 (ps), 1 (mus) or 2 (n_model row)");
    if (which == 2 && c->bb_source < 0) return fail(c, BI_ERR_INVALID, "model has no n_model tensor");
    PointGeom g;
    if (!point_geometry(c, z, g))
        return fail(c, BI_ERR_INVALID, "One of the requested xi is out of bounds");  // scipy's ValueError text
    if (which == 1) { interp_mus(c, g, out); return BI_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    const int nc = (int)g.w.size();
    const int R = which == 0 ? c->S : 1;
    std::vector<int64_t> rowoff((size_t)R * nc);
    for (int r = 0; r < R; ++r)
        for (int corner = 0; corner < nc; ++corner) {
            const int64_t a = g.cell_anchor + corner_offset(c, corner);
            rowoff[(size_t)r * nc + corner] = which == 0 ? (a * c->S + r) * c->Bp : a * c->Bp;
        }
    DevBuf d_row, d_w, d_out;
    auto cleanup = [&]() { dev_free(d_row); dev_free(d_w); dev_free(d_out); };
    if ((rc = dev_upload(c, d_row, rowoff)) || (rc = dev_upload(c, d_w, g.w)) ||
        (rc = dev_alloc(c, d_out, (size_t)R * c->B * sizeof(double)))) { cleanup(); return rc; }
    hipLaunchKernelGGL(k_morph_store, dim3((unsigned)((c->B + kThreads - 1) / kThreads), (unsigned)R), dim3(kThreads), 0,
                       c->stream, which == 0 ? (const double*)c->ps.p : (const double*)c->nm.p,
                       (const int64_t*)d_row.p, (const double*)d_w.p, nc, c->B, (double*)d_out.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out.p, (size_t)R * c->B * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_interpolate: %s", hipGetErrorString(e));
    return BI_OK;
}

int bi_eval_full(bi_ctx* c, const double* z, const double* rate_scale, int64_t dataset, double* ll, double* mus_out,
                 double* ps_out, int32_t* status) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!ll || !mus_out || !ps_out) return fail(c, BI_ERR_INVALID, "output pointers are NULL");
    if (c->bb_source >= 0 && !c->dense_counts) return fail(c, BI_ERR_STATE, "dataset counts are not resident in dense form");
    int32_t st = 0;
    rc = bi_eval(c, 1, z, rate_scale, &dataset, ll, &st);
    if (rc) return rc;
    if (status) *status = st;
    if (st & (BI_ST_OUT_OF_BOUNDS | BI_ST_UNPHYSICAL | BI_ST_BAD_DATASET)) return BI_OK;  // reference returns early
    PointGeom g;
    point_geometry(c, z, g);
    interp_mus(c, g, mus_out);
    if (rate_scale) for (int s = 0; s < c->S; ++s) mus_out[s] *= rate_scale[s];
    if ((rc = bi_interpolate(c, 0, z, ps_out))) return rc;
    if (c->bb_source < 0) return BI_OK;
    // Beeston-Barlow adjusted (mus, pmfs) for full_output (likelihood.py:656-658), on the device.
    HIP_TRY(c, hipSetDevice(c->device));
    const int i = c->bb_source;
    const int64_t B = c->B;
    double Ntot = 0.0;
    for (size_t corner = 0; corner < g.w.size(); ++corner) {
        const double term = c->h_nm_tot[(size_t)(g.cell_anchor + corner_offset(c, (int)corner))] * g.w[corner];
        Ntot = Ntot + term;
    }
    const double p_cal = mus_out[i] / Ntot;
    std::vector<double> a_row((size_t)B);
    if ((rc = bi_interpolate(c, 2, z, a_row.data()))) return rc;
    const int nblk = (int)std::min<int64_t>(1024, (B + kThreads - 1) / kThreads);
    DevBuf d_ps, d_a, d_mus, d_aw, d_part, d_tot;
    auto cleanup = [&]() { dev_free(d_ps); dev_free(d_a); dev_free(d_mus); dev_free(d_aw); dev_free(d_part); dev_free(d_tot); };
    std::vector<double> mus_v(mus_out, mus_out + c->S);
    if ((rc = dev_alloc(c, d_ps, (size_t)c->S * B * sizeof(double))) || (rc = dev_upload(c, d_a, a_row)) ||
        (rc = dev_upload(c, d_mus, mus_v)) || (rc = dev_alloc(c, d_aw, (size_t)B * sizeof(double))) ||
        (rc = dev_alloc(c, d_part, (size_t)nblk * sizeof(double))) || (rc = dev_alloc(c, d_tot, sizeof(double)))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_ps.p, ps_out, (size_t)c->S * B * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_bb_full, dim3((unsigned)nblk), dim3(kThreads), 0, c->stream, (const double*)d_ps.p,
                           (const double*)d_a.p, (const double*)c->counts.p + dataset * c->Bp, (const double*)d_mus.p,
                           c->S, i, p_cal, Ntot, B, (double*)d_aw.p, (double*)d_part.p);
        hipLaunchKernelGGL(k_rows_sum, dim3(1), dim3(64), 0, c->stream, (const double*)d_part.p, nblk, (double*)d_tot.p,
                           (int64_t)1);
        hipLaunchKernelGGL(k_bb_normalise, dim3((unsigned)((B + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                           (const double*)d_aw.p, (const double*)d_tot.p, B, (double*)d_ps.p + (size_t)i * B);
        e = hipGetLastError();
    }
    double tot = 0.0;
    if (e == hipSuccess) e = hipMemcpyAsync(ps_out + (size_t)i * B, (double*)d_ps.p + (size_t)i * B, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&tot, d_tot.p, sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_full: %s", hipGetErrorString(e));
    mus_out[i] = tot * p_cal;  // likelihood.py:658
    return BI_OK;
}

// ---- measurement ---------------------------------------------------------------------------

int bi_profile_enable(bi_ctx* c, int on) {
    if (!c) return BI_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->profiling = on != 0;
    c->ev_used = 0;
    c->prof_launches = 0;
    c->prof_ms = 0.0;
    return BI_OK;
}

int bi_profile_read(bi_ctx* c, int64_t* n_launches, double* total_ms) {
    if (!c) return BI_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double ms = 0.0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, c->ev_pool[i].first, c->ev_pool[i].second));
        ms += t;
    }
    if (n_launches) *n_launches = (int64_t)c->ev_used;
    if (total_ms) *total_ms = ms;
    c->ev_used = 0;
    return BI_OK;
}

}  // extern "C"
