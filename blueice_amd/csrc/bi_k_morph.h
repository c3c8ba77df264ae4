// bi_k_morph.h -- the morph + reduce kernels (k_morph_reduce, k_morph_single): translation unit tu_morph.hip.
#pragma once

namespace {

// The morph + reduce kernel.  blockIdx.y = item (a cell pass with up to G points),
// blockIdx.x strides over 512-bin tiles.

// The accumulate + per-bin term loop shared by the batched kernel and the single-point kernel: tiles
// tile0, tile0 + tile_step, ... of one work item.
template <int G, bool BB, bool NT, int MODE>
__device__ __forceinline__ void morph_tiles(const LaunchArgs& a, const int64_t* __restrict__ rowoff,
                                            const double* __restrict__ coef, const double* __restrict__ aux_base,
                                            const double* __restrict__ cnt, int n_tiles, int tile0, int tile_step,
                                            double (&sum)[G], unsigned (&flg)[G]) {
    // the log table travels global -> registers -> LDS; the request goes out first and lands under the first tile's
    // row loads, so a block that lives for only a few tiles does not wait for it separately
    double4 tab = {0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x < 128) tab = kLogTable[threadIdx.x];
    bool tab_pending = true;
    // XCD-aware tile order: with 8 chunks block b -- dispatched to XCD b % 8 -- streams the b % 8-th contiguous region of
    // every row instead of every 8th tile (measured +6 % on the 113-stream BB pass, +1 % on C2); short rows keep the
    // plain order, where the padded chunk count would cost some blocks a second tile
    const int chunks = (a.chunks > 1 && n_tiles >= 64 * a.chunks) ? a.chunks : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    for (int lt = tile0; lt < per_chunk * chunks; lt += tile_step) {     // (the trip count is the same for a whole block)
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double acc[G][2];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g][0] = 0.0; acc[g][1] = 0.0; }
        // the tile's counts go out FIRST: behind the stream loop their load was the one latency of a tile that nothing covered
        // (a block of the single-point call lives for two tiles: ~1 us of its 40)
        double2 nv;
        if constexpr (MODE == 2 || MODE == 3) { nv.x = nv.y = 0.0; } else { nv = *reinterpret_cast<const double2*>(cnt + bin0); }

        int k0 = 0;
        if constexpr (NT && G == 1 && MODE != 2 && MODE != 3) {
            // rows meant to stay in the Infinity Cache between calls (repeated evaluations in one cell): default policy
            k0 = a.n_keep;
#pragma unroll 8
            for (int k = 0; k < k0; ++k) {
                const double2 v = stream_load<false>(a.ps + rowoff[k] + bin0);
                const double c = coef[k];
                acc[0][0] = fma(c, v.x, acc[0][0]);
                acc[0][1] = fma(c, v.y, acc[0][1]);
            }
        }
        if constexpr (MODE == 2 || MODE == 3) if (a.nan_S > 0) {
            // np.nansum over sources (likelihood.py:686): a source whose morphed density times its rate is nan at an
            // event contributes nothing there.  Per source the corners are summed first (the morph), then the test.
            const int S = a.nan_S, nc = a.n0 / S;
            for (int s = 0; s < S; ++s) {
                double part[G][2];
#pragma unroll
                for (int g = 0; g < G; ++g) { part[g][0] = 0.0; part[g][1] = 0.0; }
                for (int c = 0; c < nc; ++c) {
                    const int k = c * S + s;
                    const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const double cf = coef[k * G + g];
                        part[g][0] = fma(cf, v.x, part[g][0]);
                        part[g][1] = fma(cf, v.y, part[g][1]);
                    }
                }
                if constexpr (MODE == 3) {        // gradient: a source dropped from the value is dropped from its slopes too
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (part[0][j] == part[0][j]) {
#pragma unroll
                            for (int g = 0; g < G; ++g) acc[g][j] += part[g][j];
                        }
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (part[g][0] == part[g][0]) acc[g][0] += part[g][0];
                        if (part[g][1] == part[g][1]) acc[g][1] += part[g][1];
                    }
                }
            }
            k0 = a.n0;
        }
#pragma unroll 8
        for (int k = k0; k < a.n0; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double c = coef[k * G + g];
                acc[g][0] = fma(c, v.x, acc[g][0]);
                acc[g][1] = fma(c, v.y, acc[g][1]);
            }
        }
        if (tab_pending) {
            if (threadIdx.x < 128) s_log_table[threadIdx.x] = tab;
            __syncthreads();
            tab_pending = false;
        }

        if constexpr (MODE == 2) {
            // extended unbinned likelihood (blueice/likelihood.py:678-690): the "bins" are the events,
            // the term is log(sum_s mu_s p_s(x_e)) with the outlier clamp; -sum_s mu_s is added by the host
            bool checked = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (a.outlier != 0.0 && !(acc[g][j] > 0.0)) acc[g][j] = a.outlier;
                    checked |= bin0 + j < a.B && !pos_normal(acc[g][j]);
                }
            }
            const bool fast = __ballot(checked) == 0ull;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double lg = fast ? bin_log_fast(acc[g][j]) : bin_log(acc[g][j]);   // (wave-uniform choice)
                    if (bin0 + j < a.B) sum[g] += lg;
                }
            }
        } else if constexpr (MODE == 3) {
            // value + gradient of the extended unbinned likelihood (blueice/likelihood.py:678-690): column 0 is the event's
            // density lambda_e = sum_s mu_s p_s(x_e), columns 1.. its derivatives; d log(lambda) = d lambda / lambda.  An event
            // that takes the outlier likelihood (lambda not > 0) is a constant: no slope.  -sum_s d mu_s is added by the host.
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double lam = acc[0][j];
                const bool clamped = a.outlier != 0.0 && !(lam > 0.0);
                if (clamped) lam = a.outlier;
                const double lg = bin_log(lam);
                if (bin0 + j < a.B) {
                    sum[0] += lg;
                    const double inv = clamped ? 0.0 : 1.0 / lam;
#pragma unroll
                    for (int g = 1; g < G; ++g) sum[g] += acc[g][j] * inv;
                }
            }
        } else if constexpr (MODE == 1) {
            sum[0] += poisson_term(nv.x, acc[0][0]) + poisson_term(nv.y, acc[0][1]);
            const double f0 = (nv.x != 0.0 ? nv.x / acc[0][0] : 0.0) - 1.0;
            const double f1 = (nv.y != 0.0 ? nv.y / acc[0][1] : 0.0) - 1.0;
#pragma unroll
            for (int g = 1; g < G; ++g) sum[g] += f0 * acc[g][0] + f1 * acc[g][1];
        } else if constexpr (!BB) {
            // per bin column of the wave: no lane has counts -> no logarithm at all (the usual case with sparse data);
            // every lane that needs one has a positive normal mu -> the branch-free form; else the checked form
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double n = j ? nv.y : nv.x;
                bool checked = false;
#pragma unroll
                for (int g = 0; g < G; ++g) checked |= needs_checked_term(n, acc[g][j]);
                if (__ballot(n > 0.0) == 0ull) {
#pragma unroll
                    for (int g = 0; g < G; ++g) sum[g] += poisson_term_nolog(n, acc[g][j]);
                } else if (__ballot(checked) == 0ull) {
#pragma unroll
                    for (int g = 0; g < G; ++g) sum[g] += poisson_term_fast(n, acc[g][j]);
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) sum[g] += poisson_term(n, acc[g][j]);
                }
            }
        } else {
            double pi[G][2], ai[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) { pi[g][0] = pi[g][1] = ai[g][0] = ai[g][1] = 0.0; }
#pragma unroll 8
            for (int k = 0; k < a.n1; ++k) {
                const double2 v = stream_load<NT>(a.ps + rowoff[a.n0 + k] + bin0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    // (the reference's own order -- value = value + V * w, blueice/pdf_morphers.py:70 via scipy -- with
                    // separate multiply and add: P_i and a feed the root formula, whose sign tests see last bits)
                    const double c = coef[(a.n0 + k) * G + g];
                    pi[g][0] = __dadd_rn(pi[g][0], __dmul_rn(v.x, c));
                    pi[g][1] = __dadd_rn(pi[g][1], __dmul_rn(v.y, c));
                }
            }
#pragma unroll 8
            for (int k = 0; k < a.n2; ++k) {
                const double2 v = stream_load<NT>(a.nm + rowoff[a.n0 + a.n1 + k] + bin0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double c = coef[(a.n0 + a.n1 + k) * G + g];
                    ai[g][0] = __dadd_rn(ai[g][0], __dmul_rn(v.x, c));
                    ai[g][1] = __dadd_rn(ai[g][1], __dmul_rn(v.y, c));
                }
            }
            const double* __restrict__ aux = aux_base;
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
                        // likelihood.py:649 asserts root1 <= 0 -- evaluated here in the reference's own operation order.
                        // (Where U_b == 0 that root is 0 analytically and its sign is decided by the last bit of the
                        // inputs; see DESIGN.md section 2 for what that means for parity.)
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

}

// MODE 2: as MODE 0 for the extended unbinned likelihood (rows hold pdf values at the events).
// MODE 3: as MODE 1 (value + gradient columns of ONE point) for the extended unbinned likelihood.
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

    morph_tiles<G, BB, NT, MODE>(a, rowoff, coef, a.aux + (int64_t)item * G * 2, cnt, n_tiles, (int)blockIdx.x, (int)gridDim.x, sum, flg);

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
    const bool fuse = a.fin_mail != nullptr;
    const int nbx = gridDim.x;
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        double s = s_sum[0][g];
        unsigned f = s_flg[0][g];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += s_sum[w][g]; f |= s_flg[w][g]; }
        const int64_t o = ((int64_t)item * nbx + blockIdx.x) * G + g;
        if (fuse) {
            if (BB) flags_post(a.fin_flags + (int64_t)item * G + g, f);
            mail_post_checked(a, a.fin_mail + o, s);
        } else {
            a.partial[o] = s;
            a.pflags[o] = f;
        }
    }
    if (!fuse || (int)blockIdx.x != nbx - 1) return;

    // ---- the item's last block does what k_finish would do, in k_finish's summation order ----
    const long long deadline = (long long)wall_clock64() + a.mail_timeout;
    double* __restrict__ mail = a.fin_mail + (int64_t)item * nbx * G;
    bool late = false;
    if (nbx <= 64) {
        // k_finish's 64-lane form: one wave per slot, lane b takes block b's partial
        for (int g = wave; g < G; g += kThreads / 64) {
            double s = lane < nbx ? mail_take(mail + (int64_t)lane * G + g, deadline, &late) : 0.0;
            s = wave_sum(s);
            const bool any_late = __ballot(late) != 0ull;
            const int64_t p = a.fin_perm[(int64_t)item * G + g];
            if (lane == 0) {
                const unsigned f = (BB ? flags_take(a.fin_flags + (int64_t)item * G + g) : 0u) | (any_late ? (unsigned)BI_ST_INTERNAL : 0u);
                if (p >= 0) {
                    a.fin_out[p] = s - a.fin_slot_lg[(int64_t)item * G + g];
                    if (a.fin_status) a.fin_status[p] |= (int32_t)f;
                }
            }
        }
        return;
    }
    // 256-lane form.  The item's slots are contiguous, [block][g]: thread t takes slots t, t + 256, ... -- always column
    // g = t % G, since G divides 256 -- four loads in flight at a time, then the threads of a column are added in a fixed
    // order (G = 1: k_finish's own order -- wave tree, then the four waves; G > 1: through LDS, thread by thread).
    {
        const int total = nbx * G;
        double s = 0.0;
        for (int q0 = threadIdx.x; q0 < total; q0 += 4 * kThreads) s = mail_take4(mail, q0, kThreads, total, s, deadline, &late);
        __shared__ double s_part[kThreads];
        __shared__ unsigned s_late[kThreads / 64];
        const unsigned lt = __ballot(late) != 0ull ? 1u : 0u;
        if constexpr (G == 1) s = wave_sum(s);
        __syncthreads();                           // (s_sum / s_flg above are done with)
        s_part[threadIdx.x] = s;
        if (lane == 0) s_late[wave] = lt;
        __syncthreads();
        if (threadIdx.x < G) {
            const int g = threadIdx.x;
            double t = 0.0;
            if constexpr (G == 1) {
                t = s_part[0];
#pragma unroll
                for (int w = 1; w < kThreads / 64; ++w) t += s_part[w * 64];
            } else {
                for (int j = 0; j < kThreads / G; ++j) t += s_part[g + G * j];
            }
            unsigned any_late = 0u;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) any_late |= s_late[w];
            const int64_t p = a.fin_perm[(int64_t)item * G + g];
            const unsigned f = (BB ? flags_take(a.fin_flags + (int64_t)item * G + g) : 0u) | (any_late ? (unsigned)BI_ST_INTERNAL : 0u);
            if (p >= 0) {
                a.fin_out[p] = t - a.fin_slot_lg[(int64_t)item * G + g];
                if (a.fin_status) a.fin_status[p] |= (int32_t)f;
            }
        }
    }
}

// ---- the single-point kernel: ONE launch from templates to scalar --------------------------------
// The call shape of `lf(**kwargs)` inside a minimizer.  The point's stream descriptors (row offsets and
// coefficients, <= kMaxSingleStreams of them) travel in the kernel-argument block, so the scalar loads hit the
// kernarg segment and no host-to-device copy precedes the launch; and the reduction is finished inside the
// launch: every block posts its partial into a mailbox slot and leaves, the last block in dispatch order collects
// them in block order (fixed order => bitwise reproducible; see mail_post) and writes {ll, status} straight into
// pinned host memory.
template <bool BB, bool NT, int MODE, bool FUSE>
__global__ __launch_bounds__(kThreads) void k_morph_single(LaunchArgs a, SingleDesc d) {
    double sum[1] = {0.0};
    unsigned flg[1] = {0u};
    morph_tiles<1, BB, NT, MODE>(a, d.rowoff, d.coef, d.aux, a.counts, a.n_tiles, (int)blockIdx.x, (int)gridDim.x, sum, flg);

    __shared__ double s_sum[kThreads / 64];
    __shared__ unsigned s_flg[kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const double s = wave_sum(sum[0]);
        const unsigned f = BB ? wave_or(flg[0]) : 0u;
        if (lane == 0) { s_sum[wave] = s; s_flg[wave] = f; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = s_sum[0];
        unsigned f = s_flg[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += s_sum[w]; f |= s_flg[w]; }
        if constexpr (FUSE) {          // post into the mailbox and leave: the last block collects (see mail_post)
            if (BB) flags_post(d.flags, f);
            mail_post_checked(a, a.partial + blockIdx.x, s);
        } else {                       // a second, tiny launch sums the partials (k_finish_single)
            a.partial[blockIdx.x] = s;
            a.pflags[blockIdx.x] = f;
        }
    }
    if constexpr (!FUSE) return;
    if (blockIdx.x != gridDim.x - 1) return;
    // the last block in dispatch order: collect the partials of all blocks in block order
    const long long deadline = (long long)wall_clock64() + a.mail_timeout;
    bool late = false;
    double s = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 4 * kThreads) s = mail_take4(a.partial, b, kThreads, (int)gridDim.x, s, deadline, &late);
    s = wave_sum(s);
    const unsigned lt = __ballot(late) != 0ull ? 1u : 0u;
    __syncthreads();   // s_sum / s_flg are reused
    if (lane == 0) { s_sum[wave] = s; s_flg[wave] = lt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_sum[0];
        unsigned any_late = s_flg[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { t += s_sum[w]; any_late |= s_flg[w]; }
        const unsigned ff = (BB ? flags_take(d.flags) : 0u) | (any_late ? (unsigned)BI_ST_INTERNAL : 0u);
        *d.out = t - d.slot_lg;
        *d.status = (int32_t)ff;
        __hip_atomic_store(d.done, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace
