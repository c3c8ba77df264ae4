// bi_k_bbgrad.h -- value + analytic gradient with Beeston-Barlow (k_morph_bbgrad): translation unit tu_grad.hip.
#pragma once

namespace {

// ---- value + analytic gradient with Beeston-Barlow (bi_eval_grad, bb_source >= 0) ----------------------------------
// mu_b = U_b + A_b p_b with p_b = r_i P_b / a_b (likelihood.py:645-646: w p_cal = P/a N * r_i/N) and A_b the second root
// of the per-bin quadratic (likelihood.py:706-708) -- or, where U_b == 0 exactly, the reference's special case
// A_b = (n_b + a_b) / (1 + p_cal) with the SCALAR p_cal = r_i / N (likelihood.py:652-653).  Everything is smooth in
// (U, P, a, r_i, N), and those are linear in the coefficient columns, so the chain rule runs per bin:
//     d mu = dU + p dA + A dp,   dA = A_a da + A_p dp + A_U dU,   dp = (dr_i P + r_i dP) / a - p da / a.
// Column 0 of the coefficient matrices gives the values (U, P, a), column q >= 1 their derivatives with respect to
// parameter q: all G columns for the U streams, but only the first DZ = 1 + d (padded) for the P and a streams -- rate
// scales do not move the Beeston-Barlow source's template or its Monte-Carlo counts -- which is what keeps the
// accumulators in registers (C5: 11 + 5 + 5 column pairs instead of 3 x 11).  aux[q] = {d r_i, d N} for q >= 1,
// aux[0] = {p_cal, N}.  The value column follows the value kernel's operation order, so ll equals bi_eval's.
template <int G, int DZ, bool NT>
__global__ __launch_bounds__(kThreads) void k_morph_bbgrad(LaunchArgs a) {
    static_assert(DZ <= G, "shape columns are a prefix of all columns");
    const int item = blockIdx.y;
    const int NS = a.n0 + a.n1 + a.n2;
    const int64_t* __restrict__ rowoff = a.rowoff + (int64_t)item * NS;
    // coefficient block of an item: [n0][G] for the U streams, then [n1][DZ] and [n2][DZ]
    const int64_t coef_per_item = (int64_t)a.n0 * G + (int64_t)(a.n1 + a.n2) * DZ;
    const double* __restrict__ cU = a.coef + (int64_t)item * coef_per_item;
    const double* __restrict__ cP = cU + (int64_t)a.n0 * G;
    const double* __restrict__ cA = cP + (int64_t)a.n1 * DZ;
    const double* __restrict__ aux = a.aux + (int64_t)item * G * 2;
    const double* __restrict__ cnt = a.counts + a.item_cnt[item];
    const int n_tiles = a.item_tiles ? a.item_tiles[item] : a.n_tiles;
    log_table_load();

    double sum[G];
    unsigned flg = 0u;
#pragma unroll
    for (int g = 0; g < G; ++g) sum[g] = 0.0;
    const double p_cal = aux[0], Ntot = aux[1];
    const double r_i = p_cal * Ntot;
    const int chunks = (a.chunks > 1 && n_tiles >= 64 * a.chunks) ? a.chunks : 1;
    const int per_chunk = (n_tiles + chunks - 1) / chunks;
    for (int lt = blockIdx.x; lt < per_chunk * chunks; lt += gridDim.x) {
        const int tile = chunks > 1 ? (lt % chunks) * per_chunk + lt / chunks : lt;
        if (tile >= n_tiles) continue;
        const int64_t bin0 = (int64_t)tile * kTile + threadIdx.x * kBinsPerThread;
        double acc[G][2], pi[DZ][2], ai[DZ][2];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g][0] = 0.0; acc[g][1] = 0.0; }
#pragma unroll
        for (int g = 0; g < DZ; ++g) { pi[g][0] = pi[g][1] = ai[g][0] = ai[g][1] = 0.0; }
#pragma unroll 4
        for (int k = 0; k < a.n0; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[k] + bin0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double c = cU[k * G + g];
                acc[g][0] = fma(c, v.x, acc[g][0]);
                acc[g][1] = fma(c, v.y, acc[g][1]);
            }
        }
#pragma unroll 4
        for (int k = 0; k < a.n1; ++k) {
            const double2 v = stream_load<NT>(a.ps + rowoff[a.n0 + k] + bin0);
            // the value column as the value kernel forms it (separate multiply and add, the reference's corner order)
            pi[0][0] = __dadd_rn(pi[0][0], __dmul_rn(v.x, cP[k * DZ]));
            pi[0][1] = __dadd_rn(pi[0][1], __dmul_rn(v.y, cP[k * DZ]));
#pragma unroll
            for (int g = 1; g < DZ; ++g) {
                const double c = cP[k * DZ + g];
                pi[g][0] = fma(c, v.x, pi[g][0]);
                pi[g][1] = fma(c, v.y, pi[g][1]);
            }
        }
#pragma unroll 4
        for (int k = 0; k < a.n2; ++k) {
            const double2 v = stream_load<NT>(a.nm + rowoff[a.n0 + a.n1 + k] + bin0);
            ai[0][0] = __dadd_rn(ai[0][0], __dmul_rn(v.x, cA[k * DZ]));
            ai[0][1] = __dadd_rn(ai[0][1], __dmul_rn(v.y, cA[k * DZ]));
#pragma unroll
            for (int g = 1; g < DZ; ++g) {
                const double c = cA[k * DZ + g];
                ai[g][0] = fma(c, v.x, ai[g][0]);
                ai[g][1] = fma(c, v.y, ai[g][1]);
            }
        }
        const double2 nv = *reinterpret_cast<const double2*>(cnt + bin0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (bin0 + j >= a.B) continue;
            const double n = j ? nv.y : nv.x;
            const double U = acc[0][j], P = pi[0][j], ab = ai[0][j];
            const double w = P / ab * Ntot;                    // likelihood.py:645-646
            const double p = w * p_cal;
            double r1, r2;
            bb_roots(ab, p, U, n, r1, r2);
            if (!(r1 <= 0.0)) flg |= BI_ST_BB_ROOT1;
            const bool special = U == 0.0;
            const double A = special ? (n + ab) / (1.0 + p_cal) : r2;
            if (!(0.0 <= A)) flg |= BI_ST_BB_NEG;
            const double mu = U + (A * w) * p_cal;
            sum[0] += poisson_term(n, mu);
            const double f = (n != 0.0 ? n / mu : 0.0) - 1.0;  // d term / d mu
            const double inv_a = 1.0 / ab;
            // partial derivatives of the second root
            double A_a = 0.0, A_p = 0.0, A_U = 0.0;
            const double t_sp = 1.0 / (1.0 + p_cal);
            if (!special) {
                const double p2 = p * p;
                const double disc = U * U * p2 + 2 * U * U * p + U * U + 2 * U * ab * p2 + 2 * U * ab * p - 2 * U * n * p2 -
                                    2 * U * n * p + ab * ab * p2 + 2 * ab * n * p2 + n * n * p2;
                const double inv_2sq = 0.5 / sqrt(disc);
                const double inv_den = 1.0 / (2 * p * (p + 1));
                const double D_a = 2 * U * p2 + 2 * U * p + 2 * ab * p2 + 2 * n * p2;
                const double D_p = 2 * U * U * p + 2 * U * U + 4 * U * ab * p + 2 * U * ab - 4 * U * n * p - 2 * U * n +
                                   2 * ab * ab * p + 4 * ab * n * p + 2 * n * n * p;
                const double D_U = 2 * U * p2 + 4 * U * p + 2 * U + 2 * ab * p2 + 2 * ab * p - 2 * n * p2 - 2 * n * p;
                A_a = (p + D_a * inv_2sq) * inv_den;
                A_U = (-p - 1.0 + D_U * inv_2sq) * inv_den;
                A_p = (-U + ab + n + D_p * inv_2sq) * inv_den - A * (4 * p + 2) * inv_den;
            }
#pragma unroll
            for (int g = 1; g < G; ++g) {
                const double dU = acc[g][j];
                const double dr = aux[g * 2 + 0];
                double dP = 0.0, da = 0.0;
                if (g < DZ) { dP = pi[g][j]; da = ai[g][j]; }
                const double dp = (dr * P + r_i * dP) * inv_a - p * da * inv_a;
                double dmu;
                if (!special) {
                    const double dA = A_a * da + A_p * dp + A_U * dU;
                    dmu = dU + p * dA + A * dp;
                } else {
                    const double dpc = (dr - p_cal * aux[g * 2 + 1]) / Ntot;
                    dmu = dp * (n + ab) * t_sp + p * da * t_sp - p * (n + ab) * t_sp * t_sp * dpc;
                }
                sum[g] += f * dmu;
            }
        }
    }

    __shared__ double s_sum[kThreads / 64][G];
    __shared__ unsigned s_flg[kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double s = wave_sum(sum[g]);
        if (lane == 0) s_sum[wave][g] = s;
    }
    {
        const unsigned f = wave_or(flg);
        if (lane == 0) s_flg[wave] = f;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        double s = s_sum[0][g];
        unsigned f = s_flg[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) { s += s_sum[w][g]; f |= s_flg[w]; }
        const int64_t o = ((int64_t)item * gridDim.x + blockIdx.x) * G + g;
        a.partial[o] = s;
        a.pflags[o] = g == 0 ? f : 0u;
    }
}

}  // namespace
