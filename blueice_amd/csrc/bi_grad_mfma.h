// bi_grad_mfma.h -- value + analytic gradient of large batches on the fp64 matrix cores (bi_eval_grad, plain binned
// likelihoods; round 4).  Host side and finish kernel: included by blueice_hip.hip after bi_planning_device.h; k_grad_mfma itself: bi_k_grad_mfma.h (tu_grad.hip).
//
// What it replaces: one work item per point (k_grad_fill + k_morph_reduce<16,...,1>), every point re-reading its cell's
// compacted rows out of L2 (2.5 MB) and carrying a [streams x 16] coefficient matrix -- 131 072 points in 40 ms, ~20 % of
// the fp64 rate.  It is the iteration of every batched profile fit (blueice_amd/profile.py; the reference's counterpart is
// scipy differencing make_objective's f, blueice/inference.py:111-124,153-155, inside the loops of :392-443).
//
// The gradient as TWO matrix products.  With mu[p][b] = sum_k coef[p][k] row[k][b] (k = corner x source) and
// d mu[p][b] / d theta_q = sum_k C_q[p][k] row[k][b]:
//     d ll / d theta_q = sum_b (n_b / mu_b - 1) d mu_b = sum_k C_q[p][k] ( G[p][k] - rowsum_k ),
//     G[p][k] = sum_b f[p][b] row[k][b],   f[p][b] = n_b / mu[p][b]   (0 where n_b = 0).
// So the device needs no derivative columns at all: per 16-bin block and 16-point work item
//     (1) mu = rows x coef         [16 bins x K] x [K x 16 points]      KG  v_mfma_f64_16x16x4
//     (2) G += f x rows^T          [16 points x 16 bins] x [16 bins x K]  4 x K/16 of them
// and the accumulator layout of (1) -- lane (kq, col) holds mu[bin 4r + kq][point col] -- IS the A-operand layout of (2)
// (row = point = col, k = kq <-> bin 4r + kq of the r-th MFMA): f goes from one product into the other without leaving its
// registers.  Only the rows are needed in two layouts (bins along the lanes for (1), streams along the lanes for (2)): two
// loads of the same cache lines.  The contraction with C_q (K x (d + S) numbers per point) is a few hundred flops per
// point and happens in the finish kernel, which rebuilds the point's geometry the way k_grad_fill does.
//
// Work decomposition: a wave owns FOUR work items of a grid cell (64 points: their coefficients and G accumulators stay in
// registers, 2 x 64 VGPRs) and one slice of the cell's 16-bin blocks; the rows of a block are loaded once for the four
// items.  Partial sums per (item, slice) are written once at the end and added in slice order by the finish kernel: no
// atomics, bitwise reproducible.
#pragma once

namespace {

// One thread per (item, slot): the slices' partial sums in slice order, then the contraction with the derivative
// coefficients C_q of the point -- d w_c / d z_i = (+-1/delta_i) prod_{j != i} w^(j), d mus_s / d z_i, as k_grad_fill
// builds them -- without ever writing the [K x (d + S)] matrix:
//     H_{c,s} = G_{c,s} - rowsum_{c,s};   A_s = sum_c w_c H_{c,s};   B_{i,s} = sum_c (d w_c / d z_i) H_{c,s}
//     d ll / d rate_scale_s = mus_s A_s;   d ll / d z_i = sum_s ( r_s B_{i,s} + (d mus_s / d z_i) rate_scale_s A_s )
// The slices' partial sums added up first, in slice order, one thread per (item, stream, point) element -- a plain streaming pass at
// full occupancy (131 072 points of C2: 200 MB in, 33 MB out) -- so that the finish kernel below, whose per-thread work arrays hold it
// to two waves per CU, reads one sixth of the data and no slice loop (it was 0.63 ms of a 4.8 ms call).  Same order of additions as the
// finish kernel's own slice loop: the same bits.
__global__ __launch_bounds__(kThreads) void k_grad_reduce_slices(const double* __restrict__ part, int64_t n_out, int64_t per_item, int n_slices,
                                                                 double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_out) return;
    const int64_t item = i / per_item, e = i - item * per_item;
    const double* __restrict__ first = part + item * n_slices * per_item + e;
    double acc = 0.0;
    int s = 0;
    for (; s + 8 <= n_slices; s += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = first[(int64_t)(s + j) * per_item];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; s < n_slices; ++s) acc += first[(int64_t)s * per_item];
    out[i] = acc;
}

constexpr int kGradFinThreads = 64;
constexpr int kGradFinDoubles = 32 + 64 + kMaxDim + kMaxDim;          // H[32], W[64], t[8], 1/delta[8]
__global__ __launch_bounds__(kGradFinThreads) void k_grad_mfma_finish(PlanMeta m, int64_t n_slots, int n_slices, int NSP,
                                                                      const int64_t* __restrict__ perm, const double* __restrict__ slot_lg,
                                                                      const double* __restrict__ part_ll, const double* __restrict__ part_g,
                                                                      const double* __restrict__ z, const double* __restrict__ rate_scale,
                                                                      double* __restrict__ ll_out, double* __restrict__ grad_out) {
    extern __shared__ double s_fin[];
    const int lane = threadIdx.x;
#define BI_AT(base, i) s_fin[((base) + (i)) * kGradFinThreads + lane]
#define H_(k) BI_AT(0, k)
#define W_(c) BI_AT(32, c)
#define T_(i) BI_AT(96, i)
#define ID_(i) BI_AT(96 + kMaxDim, i)
    const int64_t slot = (int64_t)blockIdx.x * kGradFinThreads + threadIdx.x;
    if (slot >= n_slots) return;
    const int64_t p = perm[slot];
    if (p < 0) return;
    const int64_t item = slot >> 4;
    const int gslot = (int)(slot & 15);
    const int NS = m.nc * m.S;
    // (the slices' partial sums are added in slice order, but REQUESTED eight at a time: one load in flight per thread made
    //  this kernel 180 us for 4096 points -- 16 slices x 32 streams of dependent round trips to L2)
    auto sum_slices = [&](const double* __restrict__ first, int64_t stride) {
        double acc = 0.0;
        int s = 0;
        for (; s + 8 <= n_slices; s += 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = first[(int64_t)(s + j) * stride];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; s < n_slices; ++s) acc += first[(int64_t)s * stride];
        return acc;
    };
    double ll = sum_slices(part_ll + item * n_slices * 16 + gslot, 16);
    ll -= slot_lg[slot];
    ll_out[p] = ll;
    // the point's cell and corner weights (as k_plan_geometry / k_grad_fill; the point is known to be inside the box)
    int64_t cell = 0;
    for (int i = 0; i < m.d; ++i) {
        const double* g = m.grid + m.grid_off[i];
        const int n = m.n_anchor[i];
        const double zi = z[p * m.d + i];
        int k = 0;
        double ti = 0.0, id = 0.0;
        if (n > 1) {
            if (zi == g[n - 1]) {
                k = n - 2;
            } else {
                while (k + 1 < n && g[k + 1] <= zi) ++k;
                k = min(k, n - 2);
            }
            ti = (zi - g[k]) / (g[k + 1] - g[k]);
            id = 1.0 / (g[k + 1] - g[k]);
        }
        T_(i) = ti;
        ID_(i) = id;
        cell += (int64_t)k * m.astride[i];
    }
    for (int corner = 0; corner < m.nc; ++corner) {
        double wc = 1.0;
        for (int i = 0; i < m.de; ++i) {
            const double ti = T_(m.eff_axes[i]);
            wc = wc * (((corner >> (m.de - 1 - i)) & 1) ? ti : (1 - ti));
        }
        W_(corner) = wc;
    }
    {
        const double* __restrict__ pg = part_g + (item * n_slices * NSP) * 16 + gslot;
        const int64_t stride = (int64_t)NSP * 16;
        int k = 0;
        for (; k + 2 <= NS; k += 2) {                        // (two streams together: sixteen loads in flight)
            const double g0 = sum_slices(pg + (int64_t)k * 16, stride), g1 = sum_slices(pg + (int64_t)(k + 1) * 16, stride);
            H_(k) = g0 - m.rowsum[(cell + m.corner_off[k / m.S]) * m.S + k % m.S];
            H_(k + 1) = g1 - m.rowsum[(cell + m.corner_off[(k + 1) / m.S]) * m.S + (k + 1) % m.S];
        }
        for (; k < NS; ++k) H_(k) = sum_slices(pg + (int64_t)k * 16, stride) - m.rowsum[(cell + m.corner_off[k / m.S]) * m.S + k % m.S];
    }
    auto dw = [&](int corner, int i) {
        double v = (((corner >> (m.de - 1 - i)) & 1) ? 1.0 : -1.0) * ID_(m.eff_axes[i]);
        for (int j = 0; j < m.de; ++j) {
            if (j == i) continue;
            const double tj = T_(m.eff_axes[j]);
            v *= ((corner >> (m.de - 1 - j)) & 1) ? tj : (1 - tj);
        }
        return v;
    };
    double* __restrict__ out = grad_out + p * (m.d + m.S);
    for (int i = 0; i < m.d; ++i) out[i] = 0.0;
    const bool fin = ll == ll && ll > -__builtin_inf() && ll < __builtin_inf();
    for (int src = 0; src < m.S; ++src) {
        double mus = 0.0, A = 0.0;
        for (int corner = 0; corner < m.nc; ++corner) {
            mus = mus + m.mus[(cell + m.corner_off[corner]) * m.S + src] * W_(corner);
            A += W_(corner) * H_(corner * m.S + src);
        }
        const double rs = rate_scale ? rate_scale[p * m.S + src] : 1.0;
        out[m.d + src] = fin ? mus * A : __builtin_nan("");
        for (int i = 0; i < m.de; ++i) {
            double dmus = 0.0, B = 0.0;
            for (int corner = 0; corner < m.nc; ++corner) {
                const double v = dw(corner, i);
                dmus += v * m.mus[(cell + m.corner_off[corner]) * m.S + src];
                B += v * H_(corner * m.S + src);
            }
            out[m.eff_axes[i]] += (mus * rs) * B + dmus * rs * A;
        }
    }
    if (!fin)
        for (int i = 0; i < m.d; ++i) out[i] = __builtin_nan("");
#undef BI_AT
#undef H_
#undef W_
#undef T_
#undef ID_
}

// bi_eval_grad for large single-dataset batches of a plain binned likelihood: device planner (points grouped by grid cell, 16
// per work item), k_grad_mfma, k_grad_mfma_finish.  -> ll [P], grad [P][d + S], status [P]
int eval_grad_mfma(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, bool sparse,
                   double* ll, double* grad, int32_t* status) {
    const int S = c->S, d = c->d, de = (int)c->eff_axes.size(), nc = 1 << de, NS = nc * S;
    bi_plan* plan = nullptr;
    int rc = plan_points_device(c, P, z, rate_scale, dataset, sparse, &plan, 0, 1, false, /*grad_mode=*/true);
    if (rc) return rc;
    const double ninf = -std::numeric_limits<double>::infinity(), qnan = std::numeric_limits<double>::quiet_NaN();
    std::vector<int32_t> h_st((size_t)P, 0);
    DevBuf d_pll, d_pg, d_ll, d_grad, d_rll, d_rg;
    auto cleanup = [&]() { dev_free(d_pll); dev_free(d_pg); dev_free(d_ll); dev_free(d_grad); dev_free(d_rll); dev_free(d_rg); bi_plan_destroy(c, plan); };
    hipError_t e = hipSuccess;
    bool have_results = false;                       // (results go straight into the caller's arrays: one pass over 8 P (1 + d + S) bytes less)
    if (!plan->classes.empty() && plan->classes[0].n_items > 0) {
        bi_plan::Class& k = plan->classes[0];
        const int kg = NS <= 4 ? 1 : (NS <= 8 ? 2 : (NS <= 16 ? 4 : 8));
        const int NSP = kg >= 4 ? 4 * kg : 16;
        // slices: SIX times the waves that are resident at once (two per SIMD over the whole chip), at most 16 -- cells hold
        // different numbers of points, and with one round of waves the longest quad of the fullest cell set the time (131 072
        // points of C2: kernels 6.8 ms with 1 slice, 4.9 with 2, 4.2 with 4, 4.05 with 8), while every slice adds 4 KB of
        // partial sums per work item for the finish to read (16 384 points: 1.32 ms per call with 16 slices, 2.0 with 64;
        // tools/profile/grad_slices.py); a wave wants at least ~8 blocks of 16 bins
        const int64_t quads_max = (plan->max_group_items + 3) / 4;
        const int64_t quads_all = std::max<int64_t>(1, (k.n_items + 3) / 4);
        const int64_t n_blocks = (int64_t)plan->max_item_tiles * (kTile / 16);
        const int64_t want = 6 * 2 * 4 * (int64_t)c->prop.multiProcessorCount;
        int n_slices = (int)std::max<int64_t>(1, std::min<int64_t>({(want + quads_all - 1) / quads_all, n_blocks / 8 > 0 ? n_blocks / 8 : 1, 16}));
        if (c->grad_slices > 0) n_slices = (int)std::min<int64_t>(c->grad_slices, std::max<int64_t>(1, n_blocks));
        const size_t ni = (size_t)k.n_items;
        if ((rc = dev_alloc(c, d_pll, ni * n_slices * 16 * 8)) || (rc = dev_alloc(c, d_pg, ni * n_slices * NSP * 16 * 8)) ||
            (rc = dev_alloc(c, d_ll, (size_t)P * 8)) || (rc = dev_alloc(c, d_grad, (size_t)P * (d + S) * 8))) { cleanup(); return rc; }
        GradMfmaArgs ga{};
        ga.ps = sparse ? (const double*)c->ps_c.p : (const double*)c->ps.p;
        ga.counts = sparse ? (const double*)c->cnt_c.p : (const double*)c->counts.p;
        ga.rowoff = (const int64_t*)k.rowoff.p; ga.coef = (const double*)k.coef.p;
        ga.item_cnt = (const int64_t*)k.item_cnt.p; ga.item_tiles = (const int32_t*)k.item_tiles.p;
        ga.grp_first = (const int64_t*)plan->grp_first.p; ga.grp_items = (const int32_t*)plan->grp_items.p;
        ga.part_ll = (double*)d_pll.p; ga.part_g = (double*)d_pg.p; ga.NS = NS; ga.n_slices = n_slices;
        const dim3 grid((unsigned)((quads_max * n_slices + 3) / 4), (unsigned)plan->n_groups);
        {
            EventScope ev(c);
            ++c->n_grad_mfma_launches;
            launch_grad_mfma(c, NS, grid, ga);
        }
        PlanMeta m = plan_meta_of(c, sparse);
        const int64_t n_slots = (int64_t)k.n_items * 16;
        const double* fin_ll = (const double*)d_pll.p;
        const double* fin_g = (const double*)d_pg.p;
        int fin_slices = n_slices;
        if (n_slices > 1) {
            if ((rc = dev_alloc(c, d_rll, ni * 16 * 8)) || (rc = dev_alloc(c, d_rg, ni * NSP * 16 * 8))) { (void)hipStreamSynchronize(c->stream); cleanup(); return rc; }
            const int64_t n_ll = (int64_t)ni * 16, n_g = (int64_t)ni * NSP * 16;
            hipLaunchKernelGGL(k_grad_reduce_slices, dim3((unsigned)((n_ll + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                               (const double*)d_pll.p, n_ll, (int64_t)16, n_slices, (double*)d_rll.p);
            hipLaunchKernelGGL(k_grad_reduce_slices, dim3((unsigned)((n_g + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                               (const double*)d_pg.p, n_g, (int64_t)NSP * 16, n_slices, (double*)d_rg.p);
            fin_ll = (const double*)d_rll.p;
            fin_g = (const double*)d_rg.p;
            fin_slices = 1;
        }
        const size_t lds = (size_t)kGradFinDoubles * kGradFinThreads * sizeof(double);
        e = hipFuncSetAttribute((const void*)k_grad_mfma_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_grad_mfma_finish, dim3((unsigned)((n_slots + kGradFinThreads - 1) / kGradFinThreads)), dim3(kGradFinThreads), lds,
                               c->stream, m, n_slots, fin_slices, NSP, (const int64_t*)k.perm.p, (const double*)k.slot_lg.p,
                               fin_ll, fin_g, (const double*)plan->keep_z.p,
                               rate_scale ? (const double*)plan->keep_rs.p : (const double*)nullptr, (double*)d_ll.p, (double*)d_grad.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(ll, d_ll.p, (size_t)P * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(grad, d_grad.p, (size_t)P * (d + S) * 8, hipMemcpyDeviceToHost, c->stream);
        have_results = e == hipSuccess;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_st.data(), plan->status.p, (size_t)P * 4, hipMemcpyDeviceToHost, c->stream);
    // (on an error part way the kernels already queued still write into the buffers cleanup() hands back to the recycle
    //  cache: drain the stream first, whatever it reports)
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_grad (matrix-core path): %s", hipGetErrorString(e));
    for (int64_t p = 0; p < P; ++p) {
        if (status) status[p] = h_st[(size_t)p];
        if (h_st[(size_t)p] || !have_results) {      // rejected points (and a batch without a single valid one): -inf, no slopes
            ll[p] = ninf;
            for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = qnan;
        }
    }
    return BI_OK;
}

}  // namespace
