// bi_k_grad_mfma.h -- k_grad_mfma (value + gradient of large batches on the fp64 matrix cores): translation unit
// tu_grad.hip.  The design is described at the top of bi_grad_mfma.h (host side + finish kernel, main translation unit).
#pragma once

namespace {

// Round 5.  What the kernel spent per 16-bin block of four items, beside its 64 MFMAs, was 7.4 vector instructions per MFMA on
// a chip where none of them executes beside an fp64 MFMA (40 TFLOP/s, 0.51 of the matrix peak).  Three changes:
//   * the rows are loaded ONCE: product 2 needs them with the streams along the lanes (B operand), product 1 with the bins
//     along the lanes (A operand) -- the second layout is the first one transposed through a wave-private 4 KB of LDS (rows of 17
//     doubles: conflict-free in both directions) instead of a second, gathering load of the same lines;
//   * f = n / mu for a lane's four bins from ONE reciprocal: with q = mu0 mu1 mu2 mu3 already there for the logarithm,
//     n / mu0 = mu1 (mu2 mu3) (n / q), ...: a reciprocal with two Newton steps and seven multiplications instead of four times
//     seven instructions (relative error a few ulp; the slopes are held to 1e-8);
//   * ONE logarithm per four items and block: a lane's product q_j belongs to item j's point, the four rows of the wave hold other
//     bins of the same points, so the items' products are multiplied over the rows by the transposing reduction of
//     k_scan_sorted (row j keeps item j) and every row takes the logarithm of another item;
//   and the range tests are one integer minimum over the high words (negative numbers, zeros, subnormals and nans compare low
//   or fail the window test) instead of two compares per element.
// Where the next block's rows are fetched (BI_GRAD_PREFETCH; tools/probe/grad_variants.sh, 131 072 points of C2, kernel time on one
// box): at the block's top (0) 3.55-3.61 ms; BEHIND THE BLOCK'S LAST PRODUCT-1 MFMA (2, the default) 3.47-3.55 ms -- the rows'
// registers are dead from there on, so the loads cost no register and run under the last epilogue and product 2; a whole block
// ahead (1) 4.31 ms (KG = 8 then spills 30 registers); product 1 of all four items first, then the fetch (3) 3.64-3.67 ms (20
// spills).  One wave per SIMD (BI_GRAD_WAVES = 1: 512 registers, no spills) 4.28 ms, with the fetch ahead 4.16 ms.  Nor is it its
// vector instructions: with the counts' "one value in all 16 bins" test replaced by a table looked up with a scalar load (four
// loads, eight compares and two ballots per block less: 3.5 -> ~3.0 vector instructions per MFMA) the kernel took the same time --
// and with a block's rows going from global memory straight into a double-buffered LDS tile a whole block ahead (global_load_lds_dwordx4,
// XOR-swizzled columns so that both operand layouts read it without conflicts; correct, tools/micro/global_load_lds.hip checks the
// instruction's layout) it took 3.83-3.86 ms (17-21 spilled registers).  What is left beside the matrix pipe's ~67 % is not one
// exposed latency: the vector instructions of the epilogues issue in the matrix pipe's place, and the rest is spread thin.
#ifndef BI_GRAD_PREFETCH
#define BI_GRAD_PREFETCH 2
#endif
#ifndef BI_GRAD_WAVES
#define BI_GRAD_WAVES 2
#endif
template <int KG, bool MASK>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(BI_GRAD_WAVES))) void k_grad_mfma(GradMfmaArgs a) {
    constexpr int NB = KG >= 4 ? KG / 4 : 1;       // blocks of 16 streams (product 2's N dimension)
    constexpr int kPrefetch = BI_GRAD_PREFETCH;    // 0: a block's rows at its top; 1: one block ahead; 2: behind the block's last product-1 MFMA; 3: product 1 of all four items first, then the fetch
    constexpr int NSP = 16 * NB;
    const int grp = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kq = lane >> 4, col = lane & 15;
    __shared__ int64_t s_rowoff[4 * KG];
    // the transposition buffer of each wave: [stream][bin] with rows of 17 doubles
    constexpr int kTrRow = 17;
    __shared__ double s_tr[kThreads / 64][4 * KG * kTrRow];
    if (threadIdx.x < 4 * KG) s_rowoff[threadIdx.x] = a.rowoff[a.grp_first[grp] * a.NS + min((int)threadIdx.x, a.NS - 1)];
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
    double* __restrict__ tr = s_tr[wave];

    // the four items' value coefficients: B operand of product 1 (k = kq <-> stream 4 kg + kq, column = point)
    double cf[4][KG];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double* __restrict__ c = a.coef + (item0 + min(i0 + j, n_items - 1)) * NS * 16;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) cf[j][kg] = c[min(kg * 4 + kq, NS - 1) * 16 + col];
    }
    // (the group's row offsets are read from LDS block by block: as loop invariants they would hold 2 KG registers)

    bi_double4 g[4][NB];
    double ll[4];                                  // item by item (the careful path): partial sums per lane
    double llc = 0.0;                              // the product form: lane (j, col) carries item j, point col
    bool bad[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ll[j] = 0.0;
        bad[j] = false;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) g[j][nb] = bi_double4{0.0, 0.0, 0.0, 0.0};
    }

    // (kPrefetch: the rows and counts of a block fetched one block ahead, every load unconditional -- the last block twice)
    double b1n[KG], n4n[4];
    auto fetch_rows = [&](int blk) {
        const int64_t bin0 = (int64_t)min(blk, n_blocks - 1) * 16;
        int kqo = kq;
        asm volatile("" : "+v"(kqo));                      // (opaque: keeps the LDS reads inside the loop)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) b1n[kg] = a.ps[s_rowoff[kg * 4 + kqo] + bin0 + col];
    };
    auto fetch_counts = [&](int blk) {
        const int64_t bin0 = (int64_t)min(blk, n_blocks - 1) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) n4n[r] = cnt[bin0 + 4 * r + kq];
    };
    if (kPrefetch && slice < n_blocks) fetch_rows(slice);
    if (kPrefetch == 1 && slice < n_blocks) fetch_counts(slice);
    for (int blk = slice; blk < n_blocks; blk += a.n_slices) {
        double b1[KG], b2[4][NB], n4[4];
        if (!kPrefetch) fetch_rows(blk);
        if (kPrefetch != 1) fetch_counts(blk);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) b1[kg] = (MASK && kg * 4 + kq >= NS) ? 0.0 : b1n[kg];
#pragma unroll
        for (int r = 0; r < 4; ++r) n4[r] = n4n[r];
        if (kPrefetch == 1) { fetch_rows(blk + a.n_slices); fetch_counts(blk + a.n_slices); }
        // the second layout: element (stream s, bin b) sits in lane (s & 3, b), register s >> 2; product 2 wants it in lane
        // (b & 3, s & 15), register (b >> 2, s >> 4).  Through this wave's own LDS rows (the wave's LDS operations complete in order)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) tr[(kg * 4 + kq) * kTrRow + col] = b1[kg];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                b2[r][nb] = (nb * 16 + col < 4 * KG) ? tr[min(nb * 16 + col, 4 * KG - 1) * kTrRow + 4 * r + kq] : 0.0;
        __builtin_amdgcn_wave_barrier();                   // (the next block's writes stay behind these reads)
        // the block's counts, known once for the four items: one count in all 16 bins (the rule in count order) lets a
        // lane take the product of its four expectations; anything unusual is flagged
        const double n_first = lane_value(n4[0], 0);
        const bool uniform = __builtin_amdgcn_ballot_w64(n4[0] == n_first && n4[1] == n_first && n4[2] == n_first && n4[3] == n_first) == ~0ull &&
                             n_first > 0.0 && n_first == floor(n_first);
        bool odd_lane = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) odd_lane |= n4[r] != n4[r] || n4[r] < 0.0 || n4[r] != floor(n4[r]);
        const bool odd = __builtin_amdgcn_ballot_w64(odd_lane) != 0ull;

        // item by item, scipy's values for every argument: what the product form cannot take
        auto careful = [&](int j, const bi_double4& mu, double (&f)[4]) {
            bool done = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) bad[j] |= !(mu[r] >= 0.0);
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
        // The four items in PAIRS: two chains of product 1 interleaved, the pair's f = n / mu, its product 2.  Where the block
        // takes the product form, the rule, the pair's lane products wait in qs for the one logarithm of the four items.
        double qs[4] = {1.0, 1.0, 1.0, 1.0};
        bool any_product = false;
        bi_double4 mu_all[4];
        if (kPrefetch == 3) {                              // product 1 of all four items first: the rows are dead before any epilogue
#pragma unroll
            for (int j = 0; j < 4; ++j) mu_all[j] = bi_double4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                for (int j = 0; j < 4; ++j) mu_all[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[kg], cf[j][kg], mu_all[j], 0, 0, 0);
            fetch_rows(blk + a.n_slices);
        }
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
            bi_double4 mu0 = bi_double4{0.0, 0.0, 0.0, 0.0}, mu1 = bi_double4{0.0, 0.0, 0.0, 0.0};
            if (kPrefetch == 3) {
                mu0 = mu_all[jp];
                mu1 = mu_all[jp + 1];
            } else {
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    mu0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[kg], cf[jp][kg], mu0, 0, 0, 0);
                    mu1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[kg], cf[jp + 1][kg], mu1, 0, 0, 0);
                }
            }
            // (kPrefetch == 2: the rows are dead from here on -- the next block's go into their registers while this block's
            //  last epilogue and product 2 run)
            if (kPrefetch == 2 && jp == 2) fetch_rows(blk + a.n_slices);
            // every factor above 2^-127 (one integer minimum over the eight high words: negative numbers, zeros and subnormals
            // compare low), each lane product inside (2^-255, 2^255) (nan and inf fail): the four-row products are then normal
            // numbers, and so are the reciprocals
            const int m0 = min(min(__double2hiint(mu0[0]), __double2hiint(mu0[1])), min(__double2hiint(mu0[2]), __double2hiint(mu0[3])));
            const int m1 = min(min(__double2hiint(mu1[0]), __double2hiint(mu1[1])), min(__double2hiint(mu1[2]), __double2hiint(mu1[3])));
            const double pa0 = mu0[0] * mu0[1], pb0 = mu0[2] * mu0[3], pa1 = mu1[0] * mu1[1], pb1 = mu1[2] * mu1[3];
            const double q0 = pa0 * pb0, q1 = pa1 * pb1;
            const bool ok = min(m0, m1) >= 0x38000000 && q0 > 0x1p-255 && q0 < 0x1p255 && q1 > 0x1p-255 && q1 < 0x1p255;
            double f0[4], f1[4];
            if (uniform && __builtin_amdgcn_ballot_w64(ok) == ~0ull) {
                qs[jp] = q0;
                qs[jp + 1] = q1;
                any_product = true;
                // n / q by a reciprocal refined with two Newton steps, then n / mu_r = (the other three factors) (n / q)
                double r0 = __builtin_amdgcn_rcp(q0), r1 = __builtin_amdgcn_rcp(q1);
                r0 = __builtin_fma(__builtin_fma(-q0, r0, 1.0), r0, r0);
                r1 = __builtin_fma(__builtin_fma(-q1, r1, 1.0), r1, r1);
                r0 = __builtin_fma(__builtin_fma(-q0, r0, 1.0), r0, r0);
                r1 = __builtin_fma(__builtin_fma(-q1, r1, 1.0), r1, r1);
                r0 *= n_first;
                r1 *= n_first;
                const double ta0 = pb0 * r0, tb0 = pa0 * r0, ta1 = pb1 * r1, tb1 = pa1 * r1;
                f0[0] = mu0[1] * ta0; f0[1] = mu0[0] * ta0; f0[2] = mu0[3] * tb0; f0[3] = mu0[2] * tb0;
                f1[0] = mu1[1] * ta1; f1[1] = mu1[0] * ta1; f1[2] = mu1[3] * tb1; f1[3] = mu1[2] * tb1;
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
        if (any_product) {                                 // (wave-uniform) row j <- item j's product over the four rows; one logarithm
            double x, y, P;
            BI_SWAP_MUL(__builtin_amdgcn_permlane32_swap, qs[0], qs[2], x);
            BI_SWAP_MUL(__builtin_amdgcn_permlane32_swap, qs[1], qs[3], y);
            BI_SWAP_MUL(__builtin_amdgcn_permlane16_swap, x, y, P);
            llc += n_first * bin_log_fast(P);
        }
    }

    // partial sums of this slice: ll per point (the four rows of a wave hold different bins of the same points; the product
    // form's sums sit in row j for item j), G as it lies
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j >= n_items) break;
        const int64_t slot = (item0 + i0 + j) * a.n_slices + slice;
        double t = bad[j] ? __builtin_nan("") : ll[j];
        t = rows4_sum(t);
        if (kq == j) a.part_ll[slot * 16 + col] = t + llc;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) a.part_g[(slot * NSP + nb * 16 + col) * 16 + 4 * r + kq] = g[j][nb][r];
    }
}

}  // namespace
