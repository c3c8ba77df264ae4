// bi_k_misc.h -- the small kernels of the main translation unit (blueice_hip.hip): uploads, reductions, finish kernels,
// the toy-MC form, toy generation, histogramming, event scoring.  The heavy template families live in bi_k_morph.h,
// bi_k_bbgrad.h, bi_k_scan.h, bi_scan_sorted.h and bi_grad_mfma.h, one translation unit each.
#pragma once

namespace {

// self-test hook: out[i] = bin_log(x[i])
__global__ void k_selftest_log(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
    log_table_load();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bin_log(x[i]);
}

// measurement probe: a plain sum over n2 16-byte elements -- the read-only streaming ceiling the morph kernel is
// compared with (bi_measure_read_bandwidth)
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_read_sum(const double* __restrict__ p, int64_t n2, double* __restrict__ sink) {
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 7 * stride < n2; i += 8 * stride) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = stream_load<NT>(p + 2 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) s += p[2 * i] + p[2 * i + 1];
    if (s == 0.123456789) sink[0] = s;      // keeps the loads alive, practically never stores
}

// measurement probe with the morph kernel's access pattern and nothing else: blockIdx.y = item, every item streams
// `rows` template rows of its own (row r of item i starts at element ((first + i * rows + r) % total_rows) * Bp), the
// tiles of a row are walked in the XCD-aware order of morph_tiles, 16 bytes per lane per row, `rows` loads in flight
// per lane in batches of 8 -- the ceiling `k_morph_reduce` can be held against (bi_measure_stream_bandwidth)
// PIECES: 16-byte pieces per lane per row (1 = the morph kernel's own 512-bin tiles; 2 = 1024-bin tiles, i.e. twice the
// contiguous run per row per block -- to see whether a wider tile would raise the ceiling: it does not, by more than 1 %)
template <bool NT, int PIECES>
__global__ __launch_bounds__(kThreads) void k_read_rows(const double* __restrict__ ps, int64_t Bp, int64_t total_rows,
                                                        int64_t first, int rows, int n_tiles, int chunks_in,
                                                        double* __restrict__ sink) {
    const int64_t row0 = first + (int64_t)blockIdx.y * rows;
    const int chunks = (chunks_in > 1 && n_tiles >= 64 * chunks_in) ? chunks_in : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    double s = 0.0;
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile * PIECES + threadIdx.x * kBinsPerThread;
#pragma unroll 8
        for (int r = 0; r < rows; ++r) {
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const double2 v = stream_load<NT>(ps + ((row0 + r) % total_rows) * Bp + bin0 + q * kTile);
                s += v.x + v.y;
            }
        }
    }
    if (s == 0.123456789) sink[0] = s;      // keeps the loads alive, practically never stores
}

__global__ void k_mail_init(unsigned long long* slots, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = kMailEmpty;
}

// second launch of the single-point path when the grid is too large for in-launch finishing to pay
// (an agent-scope release per block costs more than a kernel boundary once there are hundreds of blocks)
__global__ __launch_bounds__(kThreads) void k_finish_single(const double* __restrict__ partial,
                                                            const unsigned* __restrict__ pflags, int nbx, double slot_lg,
                                                            double* __restrict__ out, int32_t* __restrict__ status,
                                                            unsigned long long* done, unsigned long long seq) {
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    double s = 0.0;
    unsigned f = 0u;
#pragma unroll 8
    for (int b = threadIdx.x; b < nbx; b += kThreads) { s += partial[b]; f |= pflags[b]; }
    s = wave_sum(s);
    f = wave_or(f);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; shf[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        unsigned ff = shf[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { t += sh[w]; ff |= shf[w]; }
        *out = t - slot_lg;
        *status = (int32_t)ff;
        __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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

// OR of the per-point status words of a plan -> one word (bi_plan_status)
__global__ __launch_bounds__(kThreads) void k_status_or(const int32_t* __restrict__ status, int64_t n, int32_t* __restrict__ out) {
    unsigned f = 0u;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) f |= (unsigned)status[i];
    f = wave_or(f);
    if ((threadIdx.x & 63) == 0 && f) atomicOr((unsigned*)out, f);
}

// a resident plan's status words are OR-ed into by every run: the "gave up" bit of one run must not outlive its report
__global__ __launch_bounds__(kThreads) void k_status_clear(int32_t* __restrict__ status, int64_t n, int32_t mask) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        if (status[i] & mask) status[i] &= ~mask;
}

__global__ void k_fill_value(double* __restrict__ out, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
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
                                                          int nc, int64_t B, double* __restrict__ out,
                                                          const int32_t* __restrict__ place = nullptr) {
    const int r = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    double v = 0.0;
    for (int c = 0; c < nc; ++c) {
        const double term = __dmul_rn(src[rowoff[(int64_t)r * nc + c] + b], w[c]);
        v = __dadd_rn(v, term);
    }
    // place: column b of the tensor is the caller's column place[b] (unbinned data whose events were ordered by cell)
    out[(int64_t)r * B + (place ? (int64_t)place[b] : b)] = v;
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

// The Beeston-Barlow normalisation N(z) = n_model_events[i].sum() (likelihood.py:645) in NUMPY'S summation order, so
// that single-point calls hand bb_roots the very bits the reference computes (the knife edge at
// U_b == 0).  np.sum over a contiguous array adds, in order, the pairwise sums of 8192-element chunks (its reduction
// buffer); a chunk's pairwise sum (numpy loops_utils.h.src, pairwise_sum) splits in halves down to blocks of 128, and a
// block is summed with 8 strided accumulators r_j = a[j] + a[8 + j] + ..., combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).
// One block per chunk: interpolate a(z) for the chunk in the reference's order (value = value + V*w), reproduce that
// tree.  The (shorter) last chunk is written out as values and summed on the host by the same rule (pairwise_sum_host).
constexpr int kSumChunk = 8192;

// blockIdx.x = point (fastest in dispatch order: the points of one cell read a chunk's rows back to back, out of L2),
// blockIdx.y = chunk.  rowoff / w: [points][nc]; chunk_sum: [points][n_full]; tail: [points][B % 8192].
__global__ __launch_bounds__(kThreads) void k_bb_chunk_sums(const double* __restrict__ nm, const int64_t* __restrict__ rowoff_all,
                                                            const double* __restrict__ w_all, int nc, int64_t B,
                                                            double* __restrict__ chunk_sum_all, double* __restrict__ tail_all) {
    __shared__ double a[kSumChunk / 2];
    __shared__ double r[256];
    __shared__ double leaf[64];
    const int64_t q = blockIdx.x;
    const int64_t* __restrict__ rowoff = rowoff_all + q * nc;
    const double* __restrict__ w = w_all + q * nc;
    double* __restrict__ chunk_sum = chunk_sum_all + q * (B / kSumChunk);
    double* __restrict__ tail = tail_all + q * (B % kSumChunk);
    const int64_t b0 = (int64_t)blockIdx.y * kSumChunk;
    const int n = (int)(B - b0 < (int64_t)kSumChunk ? B - b0 : (int64_t)kSumChunk);
    if (n < kSumChunk) {                              // the last, partial chunk: values out, the host sums them
        for (int i = threadIdx.x; i < n; i += kThreads) {
            double v = 0.0;
            for (int c = 0; c < nc; ++c) v = __dadd_rn(v, __dmul_rn(nm[rowoff[c] + b0 + i], w[c]));
            tail[i] = v;
        }
        return;
    }
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        for (int i = threadIdx.x; i < kSumChunk / 2; i += kThreads) {
            double v = 0.0;
            for (int c = 0; c < nc; ++c) v = __dadd_rn(v, __dmul_rn(nm[rowoff[c] + b0 + half * (kSumChunk / 2) + i], w[c]));
            a[i] = v;
        }
        __syncthreads();
        {   // 32 blocks of 128 x 8 accumulators = 256 (block, j) pairs, one per thread
            const int blk = threadIdx.x >> 3, j = threadIdx.x & 7;
            double t = a[blk * 128 + j];
#pragma unroll
            for (int i = 1; i < 16; ++i) t = __dadd_rn(t, a[blk * 128 + 8 * i + j]);
            r[threadIdx.x] = t;
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const double* q = r + threadIdx.x * 8;
            leaf[half * 32 + threadIdx.x] = __dadd_rn(__dadd_rn(__dadd_rn(q[0], q[1]), __dadd_rn(q[2], q[3])),
                                                      __dadd_rn(__dadd_rn(q[4], q[5]), __dadd_rn(q[6], q[7])));
        }
    }
    __syncthreads();
    for (int stride = 1; stride < 64; stride <<= 1) {   // the halving recursion over 64 blocks = adjacent pairs, level by level
        if (threadIdx.x < 64 && (threadIdx.x % (2 * stride)) == 0) leaf[threadIdx.x] = __dadd_rn(leaf[threadIdx.x], leaf[threadIdx.x + stride]);
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_sum[blockIdx.y] = leaf[0];
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
    {   // (four loads in flight per thread: with one, the 37 MB tensor of a 9400-event unbinned toy took 147 us)
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        int64_t b = threadIdx.x;
        for (; b + 3 * kThreads < B; b += 4 * kThreads) {
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = r[b + k * kThreads];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s4[k] += v[k];
                mn = fmin(mn, v[k]);
                if (!(fabs(v[k]) < __builtin_inf())) bad = 1.0;
            }
        }
        for (; b < B; b += kThreads) {
            const double v = r[b];
            s4[0] += v;
            mn = fmin(mn, v);
            if (!(fabs(v) < __builtin_inf())) bad = 1.0;
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
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

// ---- toy-MC, CSR form, bin tiles staged through LDS --------------------------------------------------------------
// The row kernel above gathers log mu[idx] from an 8 MB table with one 8-byte element per cache line: at C2 (10^4
// datasets x ~9 400 non-empty bins) that gather, not HBM, sets its 0.54 ms.  Here a block owns a TILE of kDotTile bins:
// it stages that part of log mu in LDS once and walks the entries of many datasets that fall into the tile -- a
// dataset's non-empty-bin list is sorted by bin, so they are one contiguous run, found through a per-dataset table
// of tile offsets (k_csr_tile_offsets, built once per data upload).  One wave per (dataset, tile); its sum goes to
// partial[dataset][tile], which k_dataset_finish adds up in tile order: fixed order, reproducible.
constexpr int kDotTile = 8192;      // bins per tile: 64 KB of LDS

__global__ __launch_bounds__(kThreads) void k_csr_tile_offsets(const int32_t* __restrict__ nz_idx, const int64_t* __restrict__ nz_off,
                                                               int n_tl, int32_t* __restrict__ tile_off /*[T][n_tl + 1]*/,
                                                               int tile_shift /* log2 of the bins per tile */) {
    const int64_t t = blockIdx.x;
    const int64_t lo = nz_off[t], hi = nz_off[t + 1];
    int32_t* __restrict__ dst = tile_off + t * (n_tl + 1);
    if (hi == lo) {
        for (int tl = threadIdx.x; tl <= n_tl; tl += kThreads) dst[tl] = 0;
        return;
    }
    for (int64_t j = lo + threadIdx.x; j < hi; j += kThreads) {
        const int cur = nz_idx[j] >> tile_shift;
        const int prev = j > lo ? nz_idx[j - 1] >> tile_shift : -1;
        for (int tl = prev + 1; tl <= cur; ++tl) dst[tl] = (int32_t)(j - lo);     // first entry at or beyond the start of tile tl
        if (j == hi - 1)
            for (int tl = cur + 1; tl <= n_tl; ++tl) dst[tl] = (int32_t)(hi - lo);
    }
}

// sum of a double over the 16 lanes of a DPP row (lanes 16r .. 16r+15): four rotate-and-add steps on the cross-lane
// data path; every lane of the row ends up with the total
__device__ __forceinline__ double row16_sum(double v) {
#define BI_ROR_ADD(N)                                                                                              \
    do {                                                                                                           \
        const unsigned long long u = __double_as_longlong(v);                                                      \
        const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x120 + N, 0xF, 0xF, true);           \
        const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x120 + N, 0xF, 0xF, true);   \
        v += __longlong_as_double(((unsigned long long)hi << 32) | lo);                                            \
    } while (0)
    BI_ROR_ADD(8);
    BI_ROR_ADD(4);
    BI_ROR_ADD(2);
    BI_ROR_ADD(1);
#undef BI_ROR_ADD
    return v;
}

// Tile-major copy of the non-empty-bin lists: all entries of bin tile 0 (dataset 0, 1, ...), then tile 1, ... -- the
// block that owns a tile reads ONE contiguous stream instead of a sub-kilobyte run per dataset scattered over the
// dataset-major lists (which held the first tiled version at 2.6 TB/s).  An entry is 4 bytes: the byte offset of the bin
// within the staged tile (bits 3..16: bin * 8) and the count (15 bits, from bit 17), so that the kernel gets the LDS
// address with one AND and the count with one shift; data whose counts do not fit (or are not positive integers) keep
// the row kernel.  Every (dataset, tile) run is padded to a multiple of FOUR entries with kTmPadEntry -- count 0, and the
// offset of an extra LDS slot behind the tile that holds 0.0, never log mu = -inf -- so that a lane's 16-byte load is
// wholly inside its run or wholly outside it (one test per load, not per entry) and every load is 16-byte aligned.
// Round 4, measured: with the entry loads taken out the kernel runs in 24 us instead of 86 -- the ENTRY STREAM is its time
// (376 MB at C2: 54 us of HBM at best), neither the LDS gathers nor the arithmetic.  So where every count is at most 7 -- toys
// of sparse expectations: all of them -- an entry is TWO bytes: bin * 8 (the LDS byte offset, bits 3..15) | count (bits
// 0..2), eight entries per 16-byte load, runs padded to multiples of eight with 0 (count 0: the term is dropped by a select,
// there is no room for the offset of the extra slot).  Data with a larger count anywhere keeps the 4-byte entries.
constexpr uint32_t kTmPadEntry = (uint32_t)kDotTile << 3;
__global__ void k_tm_counts(const int32_t* __restrict__ tile_off, int64_t T, int n_tl, int64_t* __restrict__ cnt /*[n_tl * T + 1]*/,
                            int group /* entries per 16-byte load: 4 or 8 */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > (int64_t)n_tl * T) return;
    if (i == (int64_t)n_tl * T) { cnt[i] = 0; return; }
    const int64_t tl = i / T, t = i % T;
    const int32_t* __restrict__ o = tile_off + t * (n_tl + 1) + tl;
    cnt[i] = (o[1] - o[0] + group - 1) & ~(group - 1);     // (the run's padded length: the list is pre-filled with padding entries)
}

template <typename ENTRY>
__global__ __launch_bounds__(kThreads) void k_tm_scatter(const int32_t* __restrict__ nz_idx, const double* __restrict__ nz_n,
                                                         const int64_t* __restrict__ nz_off, const int32_t* __restrict__ tile_off,
                                                         int64_t T, int n_tl, const int64_t* __restrict__ tm_off,
                                                         ENTRY* __restrict__ tm_entries, int* __restrict__ bad, int tile_shift) {
    const int64_t t = blockIdx.x;
    const int64_t lo = nz_off[t], hi = nz_off[t + 1];
    const int32_t* __restrict__ o = tile_off + t * (n_tl + 1);
    bool any_bad = false;
    for (int64_t j = lo + threadIdx.x; j < hi; j += kThreads) {
        const int idx = nz_idx[j];
        const double n = nz_n[j];
        const int tl = idx >> tile_shift;
        if constexpr (sizeof(ENTRY) == 2) {
            if (!(n >= 1.0 && n <= 7.0 && n == floor(n))) any_bad = true;
            tm_entries[tm_off[(int64_t)tl * T + t] + (j - lo - o[tl])] = (ENTRY)(((uint32_t)(idx - (tl << tile_shift)) << 3) | ((uint32_t)n & 7u));
        } else {
            if (!(n >= 1.0 && n < 32768.0 && n == floor(n))) any_bad = true;
            tm_entries[tm_off[(int64_t)tl * T + t] + (j - lo - o[tl])] = ((uint32_t)(idx - (tl << tile_shift)) << 3) | ((uint32_t)n << 17);
        }
    }
    if (any_bad) atomicOr(bad, 1);
}

// 1024 threads: 16 waves share one staged tile; every 16-lane row of a wave takes its own dataset, so a wave has four
// (dataset, tile) runs in flight and a block 64.  The runs are short (~80 entries at C2) and each starts with a chain of
// dependent loads (offsets -> entries -> LDS): with one run after the other per row the kernel sat at 134 us for 0.4 GB
// (3 TB/s), every wave waiting a full memory latency two or three times per run.  Now a row keeps the loads of THREE
// runs in flight (ring buffers in registers, the loop unrolled over the ring so that nothing is copied and nothing
// newer than what it needs is waited for): offsets four runs ahead, entries two runs ahead.  Every load is
// unconditional -- a run behind the row's last one is the last one again (requested, never stored), an entry load
// reads up to kDotPad entries past its run (the lists are padded by that much) and its 4-entry sum is dropped afterwards
// (runs are whole 16-byte groups: round 4) -- because a branch around a load makes the compiler wait for ALL outstanding
// loads.  Index arithmetic is 32-bit, relative to the
// block's first entry (the launch keeps a block below 2^31 entries).  Per lane the entries are added in ascending
// order with fma, four at a time (one 16-byte load), the groups' sums then in ascending order: a fixed order, independent of
// the launch geometry.
typedef uint32_t bi_uint4 __attribute__((ext_vector_type(4)));
constexpr int kDotThreads = 1024;
#ifndef BI_DOT_DEPTH
#define BI_DOT_DEPTH 2
#endif
constexpr int kDotPad = 64 * 2 + 16;           // entries behind the lists that the read-ahead may touch (read, masked, never used): the largest of the variants below

// sum over the L lanes of a group inside a 16-lane DPP row (L = 16, or 8: groups start at lanes 0 and 8) on the cross-lane
// data path; every lane of the group ends up with the total.  L = 16: four rotate-and-add steps; L = 8: the mirror image
// within the half row (lane i <-> 7 - i), then the two exchanges within quads -- three steps, in a fixed order.
template <int L>
__device__ __forceinline__ double row_group_sum(double v) {
#define BI_DPP_ADD(CTRL)                                                                                           \
    do {                                                                                                           \
        const unsigned long long u = __double_as_longlong(v);                                                      \
        const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);                \
        const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);        \
        v += __longlong_as_double(((unsigned long long)hi << 32) | lo);                                            \
    } while (0)
    if constexpr (L == 16) {
        BI_DPP_ADD(0x120 + 8);        // row_ror:8
        BI_DPP_ADD(0x120 + 4);
        BI_DPP_ADD(0x120 + 2);
        BI_DPP_ADD(0x120 + 1);
    } else if constexpr (L == 8) {
        BI_DPP_ADD(0x141);            // row_half_mirror
        BI_DPP_ADD(0xB1);             // quad_perm:[1,0,3,2]
        BI_DPP_ADD(0x4E);             // quad_perm:[2,3,0,1]
    } else if constexpr (L == 4) {
        BI_DPP_ADD(0xB1);             // quad_perm:[1,0,3,2]
        BI_DPP_ADD(0x4E);             // quad_perm:[2,3,0,1]
    } else {
        static_assert(L == 2, "groups of 2, 4, 8 or 16 lanes");
        BI_DPP_ADD(0xB1);             // quad_perm:[1,0,3,2]
    }
#undef BI_DPP_ADD
    return v;
}

// L lanes per (dataset, tile) run, AHEAD 16-byte loads per lane requested up front: 4 L AHEAD entry slots per run.
// <16, 2>: 128 slots, 64 runs in flight per block (round 3).  <8, 3>: 96 slots -- a run at C2 is ~76 entries, so 79 % of the
// slots carry an entry instead of 59 % --, 128 runs in flight per block, three rotate-and-add steps instead of four (round 4).
// W: bytes per entry (4, or 2: counts up to 7) -- 16 / W entries per 16-byte load, (16 / W) L AHEAD slots per run.
template <int L, int AHEAD, int W = 4>
__global__ __launch_bounds__(kDotThreads) void k_dataset_dot_tiled(const void* __restrict__ tm_entries_v,
                                                                   const int64_t* __restrict__ tm_off, int64_t T, int n_tl,
                                                                   const double* __restrict__ logmu, int64_t B, int64_t t0,
                                                                   int64_t n, double* __restrict__ partial /*[n_tl][n]*/) {
    typedef typename std::conditional<W == 2, uint16_t, uint32_t>::type entry_t;
    constexpr int EPL = 16 / W;                                        // entries per lane and load
    static_assert(EPL * L * AHEAD + 16 <= kDotPad, "the lists' padding must cover the read-ahead");
    const entry_t* __restrict__ tm_entries = static_cast<const entry_t*>(tm_entries_v);
    __shared__ double s_mu[kDotTile + 1];                              // (+ the slot of the padding entries: 0.0)
    const int tl = blockIdx.x;
    const int64_t bin0 = (int64_t)tl * kDotTile;
    const int row = threadIdx.x / L, gl = threadIdx.x % L;             // kDotThreads / L rows of L lanes
    const int per = (int)((n + gridDim.y - 1) / gridDim.y);
    const int c0u = (int)blockIdx.y * per, c1 = min((int)n, c0u + per);
    const int c0 = min(c0u, (int)n - 1);                               // (clamped: a block without datasets still loads validly)
    const int64_t* __restrict__ off = tm_off + (int64_t)tl * T + t0;
    const int64_t base = off[c0];                                  // block-uniform: the entries of this block start here
    const entry_t* __restrict__ ent = tm_entries + base;
    const uint32_t* __restrict__ off32 = reinterpret_cast<const uint32_t*>(off);   // low words: all a block-relative index needs
    const uint32_t base32 = (uint32_t)base;
    constexpr int kStep = kDotThreads / L, kAhead = AHEAD, kDepth = BI_DOT_DEPTH, kRing = 2 * (kDepth + 1);   // entries kDepth runs ahead, offsets 2 kDepth; the ring a multiple of the entry buffers
    const int q_last = max(c0, c1 - 1);
    // (a step only ISSUES loads into the rings; whatever touches a loaded value -- the subtraction of the base, the mask
    //  of the entries past the run's end -- happens in the step that consumes it, two or four steps later: the compiler
    //  waits where a value is first used, and it does not move that use out of the step it was written in)
    auto load_offsets = [&](int q, uint32_t& ra, uint32_t& rb) {
        const int qc = min(q, q_last);
        ra = off32[2 * qc];
        rb = off32[2 * qc + 2];
    };
    // (a lane takes EPL consecutive entries per load: a group's load covers 16 L contiguous bytes of its run)
    auto load_entries = [&](uint32_t ra, bi_uint4 (&e)[kAhead]) {
        const entry_t* __restrict__ p = ent + (int)(ra - base32) + EPL * gl;
#pragma unroll
        for (int k = 0; k < kAhead; ++k) __builtin_memcpy(&e[k], p + EPL * L * k, 16);   // (16-byte load, 16-byte aligned: runs are whole groups)
    };
    bi_uint4 E[kDepth + 1][kAhead];
    uint32_t RA[kRing], RB[kRing];
    const int q0 = c0 + row;
    // the pipeline's first requests go out before the tile is staged: their latency passes under the staging
#pragma unroll
    for (int u = 0; u < 2 * kDepth; ++u) load_offsets(q0 + u * kStep, RA[u], RB[u]);
    {   // (all loads first, addresses clamped instead of predicated: eight loads in flight per thread, not one at a time)
        double v[kDotTile / kDotThreads];
#pragma unroll
        for (int k = 0; k < kDotTile / kDotThreads; ++k) v[k] = logmu[min(bin0 + threadIdx.x + k * kDotThreads, B - 1)];
#pragma unroll
        for (int k = 0; k < kDotTile / kDotThreads; ++k)
            s_mu[threadIdx.x + k * kDotThreads] = bin0 + threadIdx.x + k * kDotThreads < B ? v[k] : 0.0;
        if (threadIdx.x == 0) s_mu[kDotTile] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < kDepth; ++u) load_entries(RA[u], E[u]);
    __syncthreads();
    if (c0u >= c1) return;
    const int n_iter = (c1 - c0 + kStep - 1) / kStep;             // block-uniform: rows past their last run idle along
    for (int it = 0; it < n_iter; it += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            if (it + u < n_iter) {                               // (scalar condition)
                const int q = q0 + (it + u) * kStep;
                load_offsets(q + 2 * kDepth * kStep, RA[(u + 2 * kDepth) % kRing], RB[(u + 2 * kDepth) % kRing]);
                load_entries(RA[(u + kDepth) % kRing], E[(u + kDepth) % (kDepth + 1)]);
                __builtin_amdgcn_sched_barrier(0);               // the requests go out BEFORE this step's arithmetic, not after it
                const bi_uint4 (&e)[kAhead] = E[u % (kDepth + 1)];
                const int a = (int)(RA[u % kRing] - base32), b = (int)(RB[u % kRing] - base32);
                const int len = b - a - EPL * gl;                // (padded) entries of the run from this lane's first on
                double s = 0.0;
                if constexpr (W == 4) {
                    double lm[4 * kAhead];
#define BI_TM_LOG(x) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_mu) + ((x) & 0x1FFF8u)))
#pragma unroll
                    for (int k = 0; k < 4 * kAhead; ++k) lm[k] = BI_TM_LOG(e[k >> 2][k & 3]);      // LDS reads in flight together
#pragma unroll
                    for (int g = 0; g < kAhead; ++g) {
                        double sg = (double)(e[g][0] >> 17) * lm[4 * g];
#pragma unroll
                        for (int k = 1; k < 4; ++k) sg = __builtin_fma((double)(e[g][k] >> 17), lm[4 * g + k], sg);
                        s += 4 * L * g < len ? sg : 0.0;         // (a load behind the run's end read other runs' entries: dropped whole)
                    }
                    for (int j = a + gl + 4 * L * kAhead; j < b; j += L) {                  // (runs beyond 4 L AHEAD entries)
                        const uint32_t x = ent[j];
                        s = __builtin_fma((double)(x >> 17), BI_TM_LOG(x), s);
                    }
#undef BI_TM_LOG
                } else {
                    // two entries per word: LDS byte offset = entry & 0xFFF8, count = entry & 7 (0 = padding: its term is dropped,
                    // whatever log mu of bin 0 is)
                    double lm[8 * kAhead];
#define BI_TM_LOG16(x) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_mu) + ((x) & 0xFFF8u)))
#pragma unroll
                    for (int k = 0; k < 4 * kAhead; ++k) {
                        const uint32_t x = e[k >> 2][k & 3];
                        lm[2 * k] = BI_TM_LOG16(x);
                        lm[2 * k + 1] = BI_TM_LOG16(x >> 16);
                    }
#pragma unroll
                    for (int g = 0; g < kAhead; ++g) {
                        double sg = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t x = e[g][k];
                            const uint32_t n0 = x & 7u, n1 = (x >> 16) & 7u;
                            sg = __builtin_fma((double)n0, n0 ? lm[8 * g + 2 * k] : 0.0, sg);
                            sg = __builtin_fma((double)n1, n1 ? lm[8 * g + 2 * k + 1] : 0.0, sg);
                        }
                        s += 8 * L * g < len ? sg : 0.0;         // (a load behind the run's end read other runs' entries: dropped whole)
                    }
                    for (int j = a + gl + 8 * L * kAhead; j < b; j += L) {                  // (runs beyond 8 L AHEAD entries)
                        const uint32_t x = ent[j];
                        const uint32_t n0 = x & 7u;
                        s = __builtin_fma((double)n0, n0 ? BI_TM_LOG16(x) : 0.0, s);
                    }
#undef BI_TM_LOG16
                }
                s = row_group_sum<L>(s);
                if (gl == 0 && q < c1) partial[(int64_t)tl * n + q] = s;
            }
        }
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

// lam < 10: inversion by sequential search with one 53-bit uniform; p0 = exp(-lam) comes from a per-bin table (it is
// the same for every toy)
__device__ __forceinline__ double poisson_small(double lam, double p0, double u) {
    double p = p0, F = p, n = 0.0;
    while (u > F && n < 1000.0) {
        n += 1.0;
        p *= lam / n;
        F += p;
    }
    return n;
}

// lam >= 10: PTRS, Hoermann (1993): transformed rejection with squeeze, as in numpy's random_poisson_ptrs
__device__ double poisson_ptrs(double lam, uint64_t seed, int64_t t, int64_t b) {
    uint32_t r[4];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
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

// The draws of bins b (even) and b + 1 of toy t: one Philox block gives both their uniforms.
__device__ __forceinline__ void poisson_draw_pair(const double* __restrict__ mu, const double* __restrict__ p0, int64_t B,
                                                  uint64_t seed, int64_t t, int64_t b, double& n0, double& n1) {
    n0 = n1 = 0.0;
    if (b >= B) return;
    const double2 lam = *reinterpret_cast<const double2*>(mu + b);      // rows are padded to an even length
    const double2 e = *reinterpret_cast<const double2*>(p0 + b);
    const bool has1 = b + 1 < B;
    const bool small0 = lam.x > 0.0 && lam.x < 10.0, small1 = has1 && lam.y > 0.0 && lam.y < 10.0;
    if (small0 || small1) {
        uint32_t r[4];
        const int64_t pair = b >> 1;
        philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)t, (uint32_t)(t >> 32) & 0xFFFFu, (uint32_t)seed,
                      (uint32_t)(seed >> 32), r);
        if (small0) n0 = poisson_small(lam.x, e.x, u53(r[0], r[1]));
        if (small1) n1 = poisson_small(lam.y, e.y, u53(r[2], r[3]));
    }
    if (lam.x >= 10.0) n0 = poisson_ptrs(lam.x, seed, t, b);            // mu = 0 or invalid -> no events
    if (has1 && lam.y >= 10.0) n1 = poisson_ptrs(lam.y, seed, t, b + 1);
}

// p0[b] = exp(-mu[b]): once per bin, shared by all toys
__global__ void k_exp_neg(const double* __restrict__ mu, int64_t n, double* __restrict__ p0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p0[i] = exp(-mu[i]);
}

__global__ __launch_bounds__(kThreads) void k_toy_count(const double* __restrict__ mu, const double* __restrict__ p0, int64_t B,
                                                        uint64_t seed, int64_t t0, int32_t* __restrict__ cnt, int nchunks) {
    const int64_t t = t0 + blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    int k = 0;
#pragma unroll 1
    for (int j = 0; j < kNzPerThread; j += 2) {
        double n0, n1;
        poisson_draw_pair(mu, p0, B, seed, t, b0 + j, n0, n1);
        k += (n0 != 0.0) + (n1 != 0.0);
    }
    __shared__ int sh[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(int64_t)blockIdx.y * nchunks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(kThreads) void k_toy_scatter(const double* __restrict__ mu, const double* __restrict__ p0, int64_t B,
                                                          uint64_t seed, int64_t t0,
                                                          const int64_t* __restrict__ chunk_off, int nchunks,
                                                          int32_t* __restrict__ nz_idx, double* __restrict__ nz_n,
                                                          double* __restrict__ lg_partial) {
    const int64_t t = t0 + blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * kNzChunk + threadIdx.x * kNzPerThread;
    double v[kNzPerThread];
    int k = 0;
    double lg = 0.0;
#pragma unroll
    for (int j = 0; j < kNzPerThread; j += 2) {
        poisson_draw_pair(mu, p0, B, seed, t, b0 + j, v[j], v[j + 1]);
#pragma unroll
        for (int q = j; q < j + 2; ++q)
            if (v[q] != 0.0) { ++k; if (v[q] > 1.0) lg += lgamma(v[q] + 1.0); }
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

// ---- toy-MC generation, event by event (sparse expectations) ----------------------------------------------------
// Independent n_b ~ Poisson(mu_b) is the same law as N ~ Poisson(M = sum_b mu_b) events thrown onto the bins with
// probabilities mu_b / M.  Where M << B (10^4 expected events in 10^6 bins at C2) that is 100 times fewer random numbers
// than one draw per bin: one block per toy draws N, finds every event's bin by bisection in the cumulative sums of mu
// (8 MB, L2), sorts the bin numbers in LDS (bitonic, up to 32 768 keys = 128 KB of the CU's 160 KB) and run-length
// encodes them into the non-empty-bin list of the toy -- sorted by bin, as the per-bin generator leaves it.  Two passes
// with the same counters (Philox4x32-10 keyed by the seed; counter = (event, toy)): the first only counts the non-empty
// bins of every toy, the second writes them behind the offsets a scan made of the counts.
constexpr int kEvThreads = 512;
constexpr uint32_t kEvTag = 0x45564E54u;        // separates the event counters from the per-bin ones

__device__ __forceinline__ int toy_event_count(double M, uint64_t seed, int64_t t) {
    if (M >= 10.0) return (int)poisson_ptrs(M, seed, t, ((int64_t)1 << 40) + 7);          // a "bin" no model has
    uint32_t r[4];
    philox4x32_10(0xFFFFFFFFu, kEvTag, (uint32_t)t, (uint32_t)(t >> 32) & 0xFFFFu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (int)poisson_small(M, exp(-M), u53(r[0], r[1]));
}

// events per toy (an upper bound of its non-empty bins: the room it gets in the provisional lists)
__global__ void k_toy_event_counts(double M, uint64_t seed, int64_t t0, int64_t T, int npow2, int64_t* __restrict__ n_ev,
                                   int* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > T) return;
    if (i == T) { n_ev[i] = 0; return; }
    const int n = toy_event_count(M, seed, t0 + i);
    if (n > npow2) atomicOr(overflow, 1);
    n_ev[i] = min(n, npow2);
}

__global__ __launch_bounds__(kEvThreads) void k_toy_events(const double* __restrict__ cdf, int64_t B, double M, uint64_t seed,
                                                           int64_t t0, int npow2, const int64_t* __restrict__ room_off,
                                                           int32_t* __restrict__ idx_out, double* __restrict__ n_out,
                                                           int64_t* __restrict__ nnz_out, double* __restrict__ lgsum) {
    // LDS: radix sort (npow2 <= 16384): two key buffers of npow2 and 16 x 512 counters; bitonic sort: one key buffer;
    // then kEvThreads ints and doubles of scratch
    extern __shared__ uint32_t s_keys[];
    const bool radix = npow2 <= 16384;
    const int n_alloc = npow2;
    int key_bits = 1;
    while (key_bits < 32 && ((int64_t)1 << key_bits) < B) ++key_bits;
    int* s_scan = reinterpret_cast<int*>(s_keys + (radix ? 2 * n_alloc + 16 * kEvThreads / 2 : n_alloc));
    double* s_lg = reinterpret_cast<double*>(s_scan + kEvThreads);
    __shared__ int s_N;
    const int64_t t = t0 + blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) s_N = min(toy_event_count(M, seed, t), npow2);
    __syncthreads();
    const int N = s_N;
    int n2 = 1024;                                             // the power of two this toy needs (the sort is the cost)
    while (n2 < N) n2 <<= 1;
    for (int e = tid; e < n2; e += kEvThreads) {
        uint32_t key = 0xFFFFFFFFu;
        if (e < N) {
            uint32_t r[4];
            philox4x32_10((uint32_t)e, kEvTag, (uint32_t)t, (uint32_t)(t >> 32) & 0xFFFFu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            const double target = u53(r[0], r[1]) * M;
            int64_t lo = 0, hi = B;                                         // first bin whose cumulative sum exceeds the target
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (cdf[mid] <= target) lo = mid + 1; else hi = mid;
            }
            key = (uint32_t)min(lo, B - 1);
        }
        s_keys[e] = key;
    }
    __syncthreads();
    if (radix) {
        // LSD radix sort, 4 bits per pass, in LDS: keys ping-pong between two buffers; every thread owns a contiguous chunk
        // and a column of 16 counters, so counting and scattering need no atomics and the sort is stable.  ~100 LDS
        // accesses per thread and pass against the bitonic network's 105 stages over all keys (10^4 toys of C2: 21 -> 5 ms).
        uint32_t* src = s_keys;
        uint32_t* dst = s_keys + n_alloc;
        uint16_t* cnt = reinterpret_cast<uint16_t*>(s_keys + 2 * n_alloc);        // [16][kEvThreads]
        const int chunk = n2 / kEvThreads, c0 = tid * chunk;
        for (int shift = 0; shift < key_bits; shift += 4) {
#pragma unroll
            for (int d = 0; d < 16; ++d) cnt[d * kEvThreads + tid] = 0;
            for (int i = 0; i < chunk; ++i) ++cnt[((src[c0 + i] >> shift) & 15u) * kEvThreads + tid];
            __syncthreads();
            // exclusive scan of the 16 x 512 counters in (digit, thread) order: thread t takes elements 16 t .. 16 t + 15
            unsigned local[16], sum = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { local[q] = sum; sum += cnt[tid * 16 + q]; }
            unsigned incl = sum;
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (lane == 63) s_scan[wave] = (int)incl;
            __syncthreads();
            unsigned base = incl - sum;
            for (int w = 0; w < wave; ++w) base += (unsigned)s_scan[w];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) cnt[tid * 16 + q] = (uint16_t)(base + local[q]);
            __syncthreads();
            for (int i = 0; i < chunk; ++i) {
                const uint32_t key = src[c0 + i];
                uint16_t& slot = cnt[((key >> shift) & 15u) * kEvThreads + tid];
                dst[slot] = key;
                ++slot;
            }
            __syncthreads();
            uint32_t* t2 = src; src = dst; dst = t2;
        }
        if (src != s_keys) {                                   // an odd number of passes: bring the result home
            for (int i = tid; i < n2; i += kEvThreads) s_keys[i] = src[i];
            __syncthreads();
        }
    } else {
        for (int k = 2; k <= n2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n2; i += kEvThreads) {
                    const int x = i ^ j;
                    if (x > i) {
                        const uint32_t a = s_keys[i], b = s_keys[x];
                        if ((a > b) == ((i & k) == 0)) { s_keys[i] = b; s_keys[x] = a; }
                    }
                }
                __syncthreads();
            }
    }
    // run-length encoding: every thread owns a contiguous segment of the sorted keys
    const int seg = n2 / kEvThreads;
    const int a0 = tid * seg, a1 = a0 + seg;
    int heads = 0;
    for (int i = a0; i < a1 && i < N; ++i) heads += (i == 0 || s_keys[i] != s_keys[i - 1]);
    s_scan[tid] = heads;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < tid; ++q) base += s_scan[q];                        // (512 additions per thread: nothing beside the sort)
    int64_t pos = room_off[blockIdx.x] + base;
    double lg = 0.0;
    for (int i = a0; i < a1 && i < N; ++i) {
        if (i != 0 && s_keys[i] == s_keys[i - 1]) continue;
        int j = i + 1;
        while (j < N && s_keys[j] == s_keys[i]) ++j;
        const double n = (double)(j - i);
        idx_out[pos] = (int32_t)s_keys[i];
        n_out[pos] = n;
        ++pos;
        if (n > 1.0) lg += lgamma(n + 1.0);
    }
    s_lg[tid] = lg;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int q = 0; q < kEvThreads; ++q) tot += s_lg[q];
        lgsum[blockIdx.x] = tot;
        int nn = 0;
        for (int q = 0; q < kEvThreads; ++q) nn += s_scan[q];
        nnz_out[blockIdx.x] = nn;
    }
}

// the provisional lists (room for one entry per EVENT) packed into the final ones (one entry per non-empty bin)
__global__ __launch_bounds__(kThreads) void k_toy_pack(const int64_t* __restrict__ room_off, const int64_t* __restrict__ nz_off,
                                                       const int32_t* __restrict__ idx_in, const double* __restrict__ n_in,
                                                       int32_t* __restrict__ nz_idx, double* __restrict__ nz_n) {
    const int64_t t = blockIdx.x;
    const int64_t src = room_off[t], dst = nz_off[t], n = nz_off[t + 1] - dst;
    for (int64_t j = threadIdx.x; j < n; j += kThreads) { nz_idx[dst + j] = idx_in[src + j]; nz_n[dst + j] = n_in[src + j]; }
}

// set_data on the device: bin events into the analysis space with numpy.histogramdd semantics (what
// Histdd.add does in blueice/likelihood.py:608-609): per axis the bin is searchsorted(edges, x, 'right') - 1,
// the right-most edge is inclusive, events outside any axis range (or nan) are dropped.  Adding 1.0 with an fp64
// atomic is exact, so the result does not depend on the order of arrival.
struct HistArgs {
    int k;
    int n_edges[kMaxDim];
    int edge_off[kMaxDim];
};

__global__ __launch_bounds__(kThreads) void k_histogram(const double* __restrict__ coords /*[k][N]*/, int64_t N, HistArgs h,
                                                        const double* __restrict__ edges, double* __restrict__ counts) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    int64_t bin = 0;
    for (int a = 0; a < h.k; ++a) {
        const double x = coords[(int64_t)a * N + e];
        const double* __restrict__ g = edges + h.edge_off[a];
        const int n = h.n_edges[a];
        if (!(x >= g[0] && x <= g[n - 1])) return;      // out of range or nan
        int lo = 0, hi = n;                               // first index with g[idx] > x  (side = 'right')
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (g[mid] <= x) lo = mid + 1; else hi = mid;
        }
        int b = lo - 1;
        if (b == n - 1) b = n - 2;                        // x == last edge: belongs to the last bin
        bin = bin * (n - 1) + b;
    }
    atomicAdd(&counts[bin], 1.0);
}

// ---- unbinned set_data: histogram pdfs evaluated at the events ------------------------------------------------
// HistogramPdfSource.pdf (blueice/source.py:218-243) for every (anchor, source) row at once.  One thread = one event:
// the per-axis cell and weights are found once, then the thread walks the template rows (blockIdx.y strides over them).
//   method 0: density of the bin holding the event -- numpy.searchsorted(edges, x) - 1, clipped (nan sorts last);
//   method 1: scipy RegularGridInterpolator over the bin centres, its arithmetic: i = largest index with g[i] <= x
//             (at most n - 2), t = (x - g[i]) / (g[i+1] - g[i]); corners in itertools.product order (axis 0 slowest),
//             weight = ((1 * w_0) * w_1) ..., value = value + V * weight from 0 -- no contraction (the build sets
//             -ffp-contract=off), so the bits are scipy's.
struct ScoreArgs {
    int k, method;
    int clip;                    // 'linear': clip the coordinates to [first centre, last centre] HERE, as HistogramPdfSource.pdf does
                                 // (blueice/source.py:231-239) -- events simulated on the device keep their true coordinates
    int n_grid[kMaxDim];
    int grid_off[kMaxDim];
    int64_t stride[kMaxDim];     // bins (C order) per step along the axis
};

// Two passes (round 4).  Round 3's single kernel -- one thread per event looping over all anchors x sources rows -- let its
// blocks drift apart over the rows, so the 10^9-byte-scale gathers of 10^6 events came from a working set of gigabytes:
// every 8-byte value cost a whole memory transaction (10.6 ms for the 4 GB tensor of the C2 shape = 0.05 of the HBM peak).
// Now (1) k_score_locate finds every event's cell and interpolation weights ONCE, and (2) k_score_rows has the ROW as the
// slow grid dimension: blocks are dispatched row by row, so at any time the gathers of the whole chip fall into one or two
// 8 MB histograms that stay in L2 / Infinity Cache, and each histogram is read from HBM once (6.1 ms).  And (3) between the
// two the events are ORDERED BY CELL (radix sort of the cell indices, stable): the events of a block then fall into a few
// consecutive cache lines of the histogram, every line is fetched once per block instead of once per event, and the
// kernel becomes the stream it should be -- 4 GB in, 4 GB out.  The tensor's columns are then in sorted order; `perm`
// (sorted position -> the caller's event) stays with the context and bi_interpolate / bi_eval_full hand per-event values
// back in the caller's order, so the order is visible only to sums over events (1e-10, not bitwise, against a host-scored
// tensor).  Same arithmetic per value in the same order as before: bit-identical values.
__global__ __launch_bounds__(kThreads) void k_score_locate(const double* __restrict__ coords /*[k][N]*/, int64_t N, ScoreArgs a,
                                                           const double* __restrict__ grid, int64_t* __restrict__ base_out,
                                                           double* __restrict__ t_out /*[k][N], method 1 only*/) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    int64_t base = 0;
    for (int ax = 0; ax < a.k; ++ax) {
        double x = coords[(int64_t)ax * N + e];
        const double* __restrict__ g = grid + a.grid_off[ax];
        const int n = a.n_grid[ax];
        if (a.clip && a.method == 1) x = fmin(fmax(x, g[0]), g[n - 1]);
        int lo = 0, hi = n;
        if (a.method == 0) {                       // first index with g[idx] >= x  (side = 'left')
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (g[mid] < x) lo = mid + 1; else hi = mid;
            }
            if (x != x) lo = n;
        } else {                                   // first index with g[idx] > x
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (g[mid] <= x) lo = mid + 1; else hi = mid;
            }
        }
        const int i = min(max(lo - 1, 0), n - 2);
        if (a.method == 1) t_out[(int64_t)ax * N + e] = (x - g[i]) / (g[i + 1] - g[i]);
        base += i * a.stride[ax];
    }
    base_out[e] = base;
}

// events per thread of k_score_rows (their gathers are in flight together; one event where 2^K corners are many already)
constexpr int score_events_per_thread(int K) { return K <= 3 ? 4 : 1; }

// K: analysis dimensions of the 'linear' method (0 = 'piecewise': one value per event and row)
template <int K>
__global__ __launch_bounds__(kThreads) void k_score_rows(const int64_t* __restrict__ base_in, const double* __restrict__ t_in, int64_t N,
                                                         ScoreArgs a, const double* __restrict__ rows, int64_t row_stride, int n_rows,
                                                         double* __restrict__ out, int64_t out_stride,
                                                         const int32_t* __restrict__ perm /*NULL: base_in in event order*/) {
    constexpr int kScoreEvents = score_events_per_thread(K);
    const int64_t e0 = (int64_t)blockIdx.x * (kThreads * kScoreEvents) + threadIdx.x;
    int64_t base[kScoreEvents];
    double t[kScoreEvents][K > 0 ? K : 1];
#pragma unroll
    for (int j = 0; j < kScoreEvents; ++j) {
        const int64_t e = min(e0 + (int64_t)j * kThreads, N - 1);
        base[j] = base_in[e];                        // (sorted keys when perm is given)
        const int64_t ev = perm ? (int64_t)perm[e] : e;
#pragma unroll
        for (int ax = 0; ax < K; ++ax) t[j][ax] = t_in[(int64_t)ax * N + ev];
    }
    for (int r = blockIdx.y; r < n_rows; r += gridDim.y) {
        const double* __restrict__ row = rows + (int64_t)r * row_stride;
        double value[kScoreEvents];
        if constexpr (K == 0) {
#pragma unroll
            for (int j = 0; j < kScoreEvents; ++j) value[j] = row[base[j]];
        } else {
#pragma unroll
            for (int j = 0; j < kScoreEvents; ++j) {
                const double* __restrict__ src = row + base[j];
                double v = 0.0;
                auto add_corner = [&](int corner) {
                    double weight = 1.0;
                    int64_t off = 0;
#pragma unroll
                    for (int ax = 0; ax < K; ++ax) {
                        const bool up = (corner >> (K - 1 - ax)) & 1;
                        weight = weight * (up ? t[j][ax] : 1.0 - t[j][ax]);
                        if (up) off += a.stride[ax];
                    }
                    v = v + src[off] * weight;
                };
                if constexpr (K <= 3) {
#pragma unroll
                    for (int corner = 0; corner < (1 << K); ++corner) add_corner(corner);
                } else {
#pragma unroll 1
                    for (int corner = 0; corner < (1 << K); ++corner) add_corner(corner);
                }
                value[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < kScoreEvents; ++j) {
            const int64_t e = e0 + (int64_t)j * kThreads;
            if (e < N) out[(int64_t)r * out_stride + e] = value[j];
        }
    }
}

// ---- event-level toy Monte Carlo for histogram-pdf sources (bi_simulate_events) ---------------------------------
// Model.simulate (blueice/model.py:69-91): per source N_s ~ Poisson(mu_s) events, each drawn from the source's pdf --
// for a histogram pdf (HistogramPdfSource.simulate, source.py:248-264 -> Histdd.get_random) a bin with probability
// proportional to density x volume, then a uniform position inside the bin.  Here: pmf rows of the morphed densities and
// their running sums are prepared per source (k_sim_pmf + a scan), then one thread per event finds its source from the
// per-source counts, its bin by bisection in that source's cumulative sums and its position in the bin -- Philox4x32-10
// counters (event within its source, source) keyed by the seed, so a toy does not depend on the launch geometry.
struct SimArgs {
    int k;                       // analysis dimensions
    int S;
    int n_edges[kMaxDim];
    int edge_off[kMaxDim];
    int64_t stride[kMaxDim];     // bins (C order) per step along the axis
};
constexpr uint32_t kSimTag = 0x53494D45u;

// dens [S][B] (morphed densities) -> pmf [S][B] = density x bin volume (negative / nan densities count as 0)
__global__ __launch_bounds__(kThreads) void k_sim_pmf(const double* __restrict__ dens, SimArgs a, const double* __restrict__ edges,
                                                      int64_t B, double* __restrict__ pmf) {
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    double vol = 1.0;
    int64_t rem = b;
    for (int ax = 0; ax < a.k; ++ax) {
        const int64_t i = rem / a.stride[ax];
        rem -= i * a.stride[ax];
        const double* __restrict__ e = edges + a.edge_off[ax];
        vol *= e[i + 1] - e[i];
    }
    const double v = dens[(int64_t)blockIdx.y * B + b] * vol;
    pmf[(int64_t)blockIdx.y * B + b] = v > 0.0 ? v : 0.0;
}

// events per source: N_s ~ Poisson(rate_s), counter (source, stream)
__global__ void k_sim_counts(const double* __restrict__ rates, int S, uint64_t seed, int64_t* __restrict__ n_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double M = rates[s];
    n_out[s] = (M > 0.0 && M < 1e15) ? (int64_t)toy_event_count(M, seed ^ 0x9E3779B97F4A7C15ull, (int64_t)s) : 0;
}

__global__ __launch_bounds__(kThreads) void k_sim_events(const double* __restrict__ cdf /*[S][B]*/, int64_t B, SimArgs a,
                                                         const double* __restrict__ edges, const int64_t* __restrict__ first /*[S+1]*/,
                                                         uint64_t seed, int64_t N, double* __restrict__ coords /*[k][N]*/,
                                                         int32_t* __restrict__ source /*[N]*/) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    int s = 0;
    while (s + 1 < a.S && e >= first[s + 1]) ++s;
    const int64_t j = e - first[s];                                      // event j of source s
    const double* __restrict__ F = cdf + (int64_t)s * B;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)s, kSimTag, k0, k1, r);
    const double target = u53(r[0], r[1]) * F[B - 1];
    int64_t lo = 0, hi = B;                                              // first bin whose cumulative sum exceeds the target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (F[mid] <= target) lo = mid + 1; else hi = mid;
    }
    int64_t rem = min(lo, B - 1);
    for (int ax = 0; ax < a.k; ++ax) {
        // the position inside the bin: one uniform per axis, counter (event, source, axis)
        philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)s | ((uint32_t)ax << 24), kSimTag + 1u, k0, k1, r);
        const double u = u53(r[0], r[1]);
        const int64_t i = rem / a.stride[ax];
        rem -= i * a.stride[ax];
        const double* __restrict__ ed = edges + a.edge_off[ax];
        // (uniform inside the bin, as Histdd.get_random draws it, source.py:248-264; a 'linear' pdf clips to the outer bin
        //  centres only when it is EVALUATED, source.py:231-239: k_score_locate does that, the stored events stay as drawn)
        coords[(int64_t)ax * N + e] = ed[i] + u * (ed[i + 1] - ed[i]);
    }
    source[e] = s;
}

// small device -> pinned-host copies as a kernel (bi_memcpy_to_host): a copy-engine transfer of 80 KB costs 15 ... 110 us on
// this runtime, a launch writing through the host mapping about 10
__global__ void k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n_words) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// densify one dataset from its non-empty-bin list
__global__ void k_csr_to_dense(const int32_t* __restrict__ idx, const double* __restrict__ n, int64_t nnz,
                               double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nnz) out[idx[j]] = n[j];
}

// ---- toy-MC form: one parameter point, many datasets --------------------------------------
// pass 1: mu_b -> logmu[b] (log mu, or -inf for mu == 0, or nan for invalid mu), partial sum mu
__device__ __forceinline__ void morph_logmu_body(const LaunchArgs& a, const int64_t* __restrict__ rowoff, const double* __restrict__ coef,
                                                 double* __restrict__ logmu, int store_mu) {
    double sum = 0.0;
    unsigned bad = 0u;
    log_table_load();
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
            l.x = (m0 >= 0.0) ? bin_log(m0) : __builtin_nan("");
            l.y = (m1 >= 0.0) ? bin_log(m1) : __builtin_nan("");
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

__global__ __launch_bounds__(kThreads) void k_morph_logmu(LaunchArgs a, double* __restrict__ logmu, int store_mu) {
    morph_logmu_body(a, a.rowoff, a.coef, logmu, store_mu);
}

// The point's descriptors in the kernel arguments (up to kMaxSingleStreams streams): a toy-MC call is ONE point, and a
// host-to-device copy of its 512 bytes ahead of the launch costs more than the launch (10 ... 15 us on this runtime).
struct PointDesc {
    int64_t rowoff[kMaxSingleStreams];
    double coef[kMaxSingleStreams];
};

__global__ __launch_bounds__(kThreads) void k_morph_logmu_desc(LaunchArgs a, PointDesc d, double* __restrict__ logmu, int store_mu) {
    morph_logmu_body(a, d.rowoff, d.coef, logmu, store_mu);
}

// pass 2: for dataset t: sum_b xlogy(n_tb, mu_b).  blockIdx.y = group of kDotGroup datasets, x strides tiles: the
// log mu tile is loaded once per group (it stays in L2 / Infinity Cache: 8 MB), the counts rows stream through once
// with the nontemporal hint; XCD-aware tile order as in morph_tiles.
constexpr int kDotGroup = 8;

__device__ __forceinline__ double xlogy_term(double n, double l) {
    double t = (n > 0.0) ? n * l : 0.0;
    if (n != n) t = __builtin_nan("");
    else if (n < 0.0 || n != floor(n)) t = -__builtin_inf();
    return t;
}

__global__ __launch_bounds__(kThreads) void k_dataset_dot(const double* __restrict__ counts,
                                                          const double* __restrict__ logmu, int64_t Bp, int n_tiles,
                                                          int64_t t0, int64_t n_sets, double* __restrict__ partial) {
    const int64_t d0 = (int64_t)blockIdx.y * kDotGroup;
    const double* __restrict__ c[kDotGroup];
#pragma unroll
    for (int g = 0; g < kDotGroup; ++g) c[g] = counts + (t0 + min(d0 + g, n_sets - 1)) * Bp;   // tail group: repeats the last row
    double s[kDotGroup];
#pragma unroll
    for (int g = 0; g < kDotGroup; ++g) s[g] = 0.0;
    const int chunks = n_tiles >= 64 * 8 ? 8 : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double2 n[kDotGroup];
#pragma unroll
        for (int g = 0; g < kDotGroup; ++g) n[g] = stream_load<true>(c[g] + bin0);
        const double2 l = *reinterpret_cast<const double2*>(logmu + bin0);
#pragma unroll
        for (int g = 0; g < kDotGroup; ++g) s[g] += xlogy_term(n[g].x, l.x) + xlogy_term(n[g].y, l.y);
    }
    __shared__ double sh[kThreads / 64][kDotGroup];
#pragma unroll
    for (int g = 0; g < kDotGroup; ++g) {
        const double w = wave_sum(s[g]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][g] = w;
    }
    __syncthreads();
    if (threadIdx.x < kDotGroup && d0 + threadIdx.x < n_sets) {
        double t = sh[0][threadIdx.x];
        for (int w = 1; w < kThreads / 64; ++w) t += sh[w][threadIdx.x];
        partial[(d0 + threadIdx.x) * gridDim.x + blockIdx.x] = t;
    }
}

// out[t] = sum_blocks partial[t][:] - summu - lgsum[t0 + t]   (nan if any mu invalid).  256 threads per block: the
// block first sums the mu partials of pass 1 together (fixed tree), then every thread finishes one dataset.
// (partial[t * t_stride + b * b_stride]: dataset-major from the row kernels, block-major from the tiled one, whose 123
// partials per dataset would otherwise be read with a stride of 123 doubles between neighbouring threads)
__global__ __launch_bounds__(kThreads) void k_dataset_finish(const double* __restrict__ partial, int nbx, int64_t t_stride, int64_t b_stride,
                                                             const double* __restrict__ mu_partial,
                                                             const unsigned* __restrict__ mu_flags, int nmu,
                                                             const double* __restrict__ lgsum, int64_t t0, int64_t n,
                                                             double* __restrict__ out) {
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    double m = 0.0;
    unsigned f = 0u;
    {   // (four loads in flight per thread: every block of this kernel repeats this sum before it can start on its datasets)
        double m4[4] = {0.0, 0.0, 0.0, 0.0};
        int b = threadIdx.x;
        for (; b + 3 * kThreads < nmu; b += 4 * kThreads) {
            double v[4];
            unsigned g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = mu_partial[b + k * kThreads]; g[k] = mu_flags[b + k * kThreads]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { m4[k] += v[k]; f |= g[k]; }
        }
        for (; b < nmu; b += kThreads) { m4[0] += mu_partial[b]; f |= mu_flags[b]; }
        m = (m4[0] + m4[1]) + (m4[2] + m4[3]);
    }
    m = wave_sum(m);
    f = wave_or(f);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = m; shf[threadIdx.x >> 6] = f; }
    __syncthreads();
    m = sh[0];
    f = shf[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) { m += sh[w]; f |= shf[w]; }
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n) return;
    // eight running sums: the loads of a thread do not wait for one another (123 block-major partials per dataset from
    // the tiled kernel are 123 trips to L2 otherwise); fixed order all the same
    const double* __restrict__ p = partial + t * t_stride;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int b = 0;
    for (; b + 7 < nbx; b += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(int64_t)(b + k) * b_stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    for (; b < nbx; ++b) acc[0] += p[(int64_t)b * b_stride];
    const double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    double r = (s - m) - lgsum[t0 + t];
    if (f) r = __builtin_nan("");
    out[t] = r;
}

// The finish of the tiled toy-MC pass, partial[tile][dataset] (block-major): 64 datasets per block, the tiles of a
// dataset split over the block's four waves (k_dataset_finish has one thread walk all ~123 tiles of its dataset, 16 trips
// to L2 one after the other on 40 blocks: 10 ... 15 us for 10^4 datasets; here 157 blocks and four trips).  Fixed order:
// a wave's range in steps of eight with eight running sums, then the four waves' sums ((0 + 1) + (2 + 3)).
// done != NULL: out is pinned host memory; the block that finishes last publishes `seq` there with a system-scope
// release once every block's results are on their way (the host polls the word instead of synchronising the stream).
__global__ __launch_bounds__(kThreads) void k_dataset_finish_tiled(const double* __restrict__ partial, int nbx,
                                                                   const double* __restrict__ mu_partial,
                                                                   const unsigned* __restrict__ mu_flags, int nmu,
                                                                   const double* __restrict__ lgsum, int64_t t0, int64_t n,
                                                                   double* __restrict__ out, unsigned* __restrict__ blocks_done,
                                                                   unsigned long long* done, unsigned long long seq) {
    static_assert(kThreads == 256, "four waves per block");
    __shared__ double sh[kThreads / 64];
    __shared__ unsigned shf[kThreads / 64];
    __shared__ double part[kThreads / 64][64];
    double m = 0.0;
    unsigned f = 0u;
    {
        double m4[4] = {0.0, 0.0, 0.0, 0.0};
        int b = threadIdx.x;
        for (; b + 3 * kThreads < nmu; b += 4 * kThreads) {
            double v[4];
            unsigned g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = mu_partial[b + k * kThreads]; g[k] = mu_flags[b + k * kThreads]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { m4[k] += v[k]; f |= g[k]; }
        }
        for (; b < nmu; b += kThreads) { m4[0] += mu_partial[b]; f |= mu_flags[b]; }
        m = (m4[0] + m4[1]) + (m4[2] + m4[3]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 64 + lane;
    const int per = (nbx + 3) / 4;
    const int b0 = wave * per, b1 = min(nbx, b0 + per);
    double s = 0.0;
    if (t < n) {
        const double* __restrict__ p = partial + t;
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int b = b0;
        for (; b + 7 < b1; b += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(int64_t)(b + k) * n];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
        for (; b < b1; ++b) acc[0] += p[(int64_t)b * n];
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    part[wave][lane] = s;
    m = wave_sum(m);
    f = wave_or(f);
    if (lane == 0) { sh[wave] = m; shf[wave] = f; }
    __syncthreads();
    if (wave == 0 && t < n) {
        double mm = sh[0];
        unsigned ff = shf[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { mm += sh[w]; ff |= shf[w]; }
        const double tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        double r = (tot - mm) - lgsum[t0 + t];
        if (ff) r = __builtin_nan("");
        out[t] = r;
    }
    if (!done) return;
    if (wave == 0) {
        __threadfence_system();                        // this wave's results have left before the block is counted
        if (lane == 0) {
            const unsigned before = __hip_atomic_fetch_add(blocks_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before == gridDim.x - 1) {
                __hip_atomic_store(blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (ready for the next call)
                __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// The finish of a scan plan: partial is [items][nslots][16] (point fastest), one wave per item.  Lane l takes point
// l & 15 and the slots l >> 4, l >> 4 + 4, ...: every load instruction of the wave covers 64 consecutive doubles (k_finish
// walks the same array with one 8-byte element per 128-byte line and took 1.3 ms for the 640 MB of a 10^6-point scan;
// this takes 0.1).  Fixed order: slots in steps of four per lane, then the four lane groups -- bitwise reproducible.
__global__ __launch_bounds__(kThreads) void k_finish_scan(const double* __restrict__ partial, int nslots, int64_t n_items,
                                                          const int64_t* __restrict__ perm, const double* __restrict__ slot_lg,
                                                          double* __restrict__ out) {
    const int64_t item = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int lane = threadIdx.x & 63, g = lane & 15;
    const double* __restrict__ src = partial + item * nslots * 16 + g;
    double s = 0.0;
#pragma unroll 4
    for (int b = lane >> 4; b < nslots; b += 4) s += src[(int64_t)b * 16];
    s = rows4_sum(s);
    if (lane < 16) {
        const int64_t p = perm[item * 16 + g];
        if (p >= 0) out[p] = s - slot_lg[item * 16 + g];
    }
}

// out[perm[slot]] = nan where the validity pass flagged the slot
__global__ __launch_bounds__(kThreads) void k_apply_bad(const unsigned* __restrict__ bad, const int64_t* __restrict__ perm,
                                                        int64_t n_slots, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_slots || !bad[i]) return;
    const int64_t p = perm[i];
    if (p >= 0) out[p] = __builtin_nan("");
}

}  // namespace
