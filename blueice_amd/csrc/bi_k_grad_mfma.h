// bi_k_grad_mfma.h -- k_grad_mfma (value + gradient of large batches on the fp64 matrix cores): translation unit
// tu_grad.hip.  The design is described at the top of bi_grad_mfma.h (host side + finish kernel, main translation unit).
#pragma once

namespace {

template <int KG, bool MASK>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2))) void k_grad_mfma(GradMfmaArgs a) {
    constexpr int NB = KG >= 4 ? KG / 4 : 1;       // blocks of 16 streams (product 2's N dimension)
    constexpr int NSP = 16 * NB;
    const int grp = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kq = lane >> 4, col = lane & 15;
    // (product 2 reads the rows of streams 0 .. 16 NB - 1 -- more than the 4 KG of product 1 when KG < 4: every index is
    //  clamped to a valid row here, and the rows of streams beyond NS are zeroed where they are loaded)
    constexpr int NRO = 4 * KG > NSP ? 4 * KG : NSP;
    constexpr bool MASK_B2 = MASK || NSP != 4 * KG;
    __shared__ int64_t s_rowoff[NRO];
    if (threadIdx.x < NRO) s_rowoff[threadIdx.x] = a.rowoff[a.grp_first[grp] * a.NS + min((int)threadIdx.x, a.NS - 1)];
    log_table_load();
    const int wx = blockIdx.x * 4 + wave;
    const int quad = wx / a.n_slices, slice = wx % a.n_slices;
    const int64_t item0 = a.grp_first[grp];
    const int n_items = a.grp_items[grp];
    const int i0 = quad * 4;
    if (i0 >= n_items) return;
    const int NS = a.NS;
    const double* __restrict__ cnt = a.counts + a.item_cnt[item0];
    const int n_blocks = a.item_tiles[item0] * (kTile / 16);

    // the four items' value coefficients: B operand of product 1 (k = kq <-> stream 4 kg + kq, column = point)
    double cf[4][KG];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double* __restrict__ c = a.coef + (item0 + min(i0 + j, n_items - 1)) * NS * 16;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) cf[j][kg] = c[min(kg * 4 + kq, NS - 1) * 16 + col];
    }
    // (the group's row offsets are read from LDS block by block: as loop invariants they would hold 2 (KG + NB) registers)

    bi_double4 g[4][NB];
    double ll[4];
    bool bad[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ll[j] = 0.0;
        bad[j] = false;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) g[j][nb] = bi_double4{0.0, 0.0, 0.0, 0.0};
    }

    for (int blk = slice; blk < n_blocks; blk += a.n_slices) {
        const int64_t bin0 = (int64_t)blk * 16;
        double b1[KG], b2[4][NB], n4[4];
        int kqo = kq, colo = col;
        asm volatile("" : "+v"(kqo), "+v"(colo));          // (opaque: keeps the LDS reads inside the loop)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const double v = a.ps[s_rowoff[kg * 4 + kqo] + bin0 + col];
            b1[kg] = (MASK && kg * 4 + kq >= NS) ? 0.0 : v;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int64_t row = s_rowoff[nb * 16 + colo];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = a.ps[row + bin0 + 4 * r + kq];
                b2[r][nb] = (MASK_B2 && nb * 16 + col >= NS) ? 0.0 : v;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) n4[r] = cnt[bin0 + 4 * r + kq];
        // the block's counts, known once for the four items: one count in all 16 bins (the rule in count order) lets a
        // lane take ONE logarithm of the product of its four expectations; anything unusual is flagged
        const double n_first = lane_value(n4[0], 0);
        const bool uniform = __builtin_amdgcn_ballot_w64(n4[0] == n_first && n4[1] == n_first && n4[2] == n_first && n4[3] == n_first) == ~0ull &&
                             n_first > 0.0 && n_first == floor(n_first);
        bool odd_lane = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) odd_lane |= n4[r] != n4[r] || n4[r] < 0.0 || n4[r] != floor(n4[r]);
        const bool odd = __builtin_amdgcn_ballot_w64(odd_lane) != 0ull;

        // The four items in PAIRS: two chains of product 1 interleaved, then -- where the block takes the product form for
        // both items, the rule -- ONE straight-line epilogue for the two (two logarithms and eight reciprocals in flight
        // together: with two waves per SIMD a single item's dependent chains leave the vector unit waiting on itself), then
        // the two items' product 2.  Anything unusual falls back to the item-by-item epilogue with scipy's values for
        // every argument.  (9 % of the kernel at 131 072 points.)
        auto careful = [&](int j, const bi_double4& mu, double (&f)[4]) {
            bool done = false;
            if (uniform) {
                const bool ok = mu[0] > kProdFloor && mu[1] > kProdFloor && mu[2] > kProdFloor && mu[3] > kProdFloor;
                const double q = (mu[0] * mu[1]) * (mu[2] * mu[3]);
                if (__builtin_amdgcn_ballot_w64(ok && pos_normal(q)) == ~0ull) {
                    ll[j] += n_first * bin_log_fast(q);
                    done = true;
                }
            }
            if (!done) {
                bool checked = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) checked |= n4[r] > 0.0 && !pos_normal(mu[r]);
                if (__builtin_amdgcn_ballot_w64(checked) == 0ull) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double lg = bin_log_fast(mu[r]);
                        ll[j] += n4[r] > 0.0 ? n4[r] * lg : 0.0;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ll[j] += n4[r] > 0.0 ? n4[r] * bin_log(mu[r]) : 0.0;
                }
                if (odd) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n4[r] != n4[r]) bad[j] = true;
                        else if (n4[r] < 0.0 || n4[r] != floor(n4[r])) ll[j] += -__builtin_inf();
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) f[r] = n4[r] != 0.0 ? n4[r] / mu[r] : 0.0;
        };
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
            bi_double4 mu0 = bi_double4{0.0, 0.0, 0.0, 0.0}, mu1 = bi_double4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                mu0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[kg], cf[jp][kg], mu0, 0, 0, 0);
                mu1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[kg], cf[jp + 1][kg], mu1, 0, 0, 0);
            }
            // sum_b n log mu over the lane's four bins of its point, and f = n / mu for product 2
            bool neg0 = false, neg1 = false, ok = uniform;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                neg0 |= !(mu0[r] >= 0.0);
                neg1 |= !(mu1[r] >= 0.0);
                ok &= mu0[r] > kProdFloor && mu1[r] > kProdFloor;
            }
            bad[jp] |= neg0;
            bad[jp + 1] |= neg1;
            const double q0 = (mu0[0] * mu0[1]) * (mu0[2] * mu0[3]), q1 = (mu1[0] * mu1[1]) * (mu1[2] * mu1[3]);
            double f0[4], f1[4];
            if (__builtin_amdgcn_ballot_w64(ok && pos_normal(q0) && pos_normal(q1)) == ~0ull) {
                // (one positive count, every mu a normal number above 2^-127: n / mu as n times a reciprocal refined by two
                //  Newton steps -- relative error < 2^-50, the slopes are held to 1e-8 --, 7 instructions instead of 15)
                ll[jp] += n_first * bin_log_fast(q0);
                ll[jp + 1] += n_first * bin_log_fast(q1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double r0 = __builtin_amdgcn_rcp(mu0[r]), r1 = __builtin_amdgcn_rcp(mu1[r]);
                    r0 = __builtin_fma(__builtin_fma(-mu0[r], r0, 1.0), r0, r0);
                    r1 = __builtin_fma(__builtin_fma(-mu1[r], r1, 1.0), r1, r1);
                    r0 = __builtin_fma(__builtin_fma(-mu0[r], r0, 1.0), r0, r0);
                    r1 = __builtin_fma(__builtin_fma(-mu1[r], r1, 1.0), r1, r1);
                    f0[r] = n_first * r0;
                    f1[r] = n_first * r1;
                }
            } else {
                careful(jp, mu0, f0);
                careful(jp + 1, mu1, f1);
            }
            // f feeds product 2 straight from these registers: A[m = point = col][k = kq <-> bin 4 r + kq]
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    g[jp][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[r], b2[r][nb], g[jp][nb], 0, 0, 0);
                    g[jp + 1][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[r], b2[r][nb], g[jp + 1][nb], 0, 0, 0);
                }
        }
    }

    // partial sums of this slice: ll per point (the four rows of a wave hold different bins of the same points), G as it lies
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j >= n_items) break;
        const int64_t slot = (item0 + i0 + j) * a.n_slices + slice;
        double t = bad[j] ? __builtin_nan("") : ll[j];
        t = rows4_sum(t);
        if (kq == 0) a.part_ll[slot * 16 + col] = t;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) a.part_g[(slot * NSP + nb * 16 + col) * 16 + 4 * r + kq] = g[j][nb][r];
    }
}

}  // namespace
