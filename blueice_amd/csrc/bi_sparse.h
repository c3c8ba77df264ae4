// bi_sparse.h -- non-empty-bin forms of the data: CSR lists and per-dataset compacted templates.
#pragma once

#include <rocprim/rocprim.hpp>

namespace {

// per-dataset compacted copies of all template rows over the non-empty bins (needs the CSR lists)
int build_compact_templates(bi_ctx* c) {
    c->compact_ready = false;
    const int64_t T = c->T, Bp = c->Bp;
    int rc;
    hipError_t e;
    if (c->bb_source >= 0 || !c->ps_finite) return BI_OK;   // (templates of either sign: the identity holds, only the validity of empty bins does not follow -- see k_scan_valid)
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
    // (the row totals of ALL datasets land in one scratch block and come back in one copy: a synchronisation per dataset
    //  was most of the time of compacting a few hundred toys)
    // (when the compacted copy has to grow it grows by an eighth more than asked: the next toy ensemble is a few tiles larger
    //  or smaller, and giving 10 GB back to the driver and asking for 10 GB + 4 MB costs 0.4 s)
    auto roomy = [&](DevBuf& b, size_t bytes) { return b.p && b.bytes >= bytes ? BI_OK : dev_alloc(c, b, bytes + bytes / 8); };
    if ((rc = roomy(c->ps_c, (size_t)tot_ps * sizeof(double))) || (rc = roomy(c->cnt_c, (size_t)tot_cnt * sizeof(double))) ||
        (rc = dev_alloc(c, c->scratch, (size_t)T * rows * sizeof(double))))
        return rc;
    c->h_Tz.assign((size_t)T * rows, 0.0);
    std::vector<double> tnz((size_t)T * rows);
    // Few datasets (a scan's one): the compacted copy holds the non-empty bins ORDERED BY THEIR COUNT (ties by bin) -- every
    // consumer of the copy sums over its bins, so the order is free, and the matrix-core scan kernel turns runs of equal
    // counts into one logarithm per lane and strip (k_scan_mfma PROD = 2).  The CSR lists themselves stay in bin order.
    c->compact_sorted = false;
    const bool sort_by_count = c->scan_pow && T <= 64;
    DevBuf d_sidx, d_sn, d_tmp;
    auto drop = [&]() { dev_free(d_sidx); dev_free(d_sn); dev_free(d_tmp); };
    for (int64_t t = 0; t < T; ++t) {
        const int64_t lo = c->h_nz_off[(size_t)t], nnz = c->h_nz_off[(size_t)t + 1] - lo, np = c->h_c_np[(size_t)t];
        double* dst = (double*)c->ps_c.p + c->h_c_off[(size_t)t];
        const int32_t* idx = (const int32_t*)c->nz_idx.p + lo;
        const double* cnt = (const double*)c->nz_n.p + lo;
        if (sort_by_count && nnz > 1) {
            size_t tmp_bytes = 0;
            (void)prim_sort_pairs(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr,
                                            (int32_t*)nullptr, (size_t)nnz, 0u, 64u, c->stream);
            if ((rc = dev_alloc(c, d_sidx, (size_t)nnz * sizeof(int32_t))) || (rc = dev_alloc(c, d_sn, (size_t)nnz * sizeof(double))) ||
                (rc = dev_alloc(c, d_tmp, std::max<size_t>(tmp_bytes, 256)))) { drop(); return rc; }
            size_t tb = d_tmp.bytes;
            e = prim_sort_pairs(d_tmp.p, tb, cnt, (double*)d_sn.p, idx, (int32_t*)d_sidx.p, (size_t)nnz, 0u, 64u, c->stream);
            if (e != hipSuccess) { drop(); return fail(c, BI_ERR_HIP, "template compaction (sort by count): %s", hipGetErrorString(e)); }
            idx = (const int32_t*)d_sidx.p;
            cnt = (const double*)d_sn.p;
        }
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((np + kThreads - 1) / kThreads), (unsigned)rows), dim3(kThreads), 0,
                           c->stream, (const double*)c->ps.p, Bp, idx, nnz, np, dst);
        hipLaunchKernelGGL(k_pad_copy, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream,
                           cnt, nnz, np, (double*)c->cnt_c.p + c->h_cnt_off[(size_t)t]);
        hipLaunchKernelGGL(k_row_total, dim3((unsigned)rows), dim3(kThreads), 0, c->stream, (const double*)dst, np, np,
                           (double*)c->scratch.p + t * rows);
        e = hipGetLastError();
        if (e != hipSuccess) { (void)hipStreamSynchronize(c->stream); drop(); return fail(c, BI_ERR_HIP, "template compaction: %s", hipGetErrorString(e)); }
    }
    e = hipMemcpyAsync(tnz.data(), c->scratch.p, tnz.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "template compaction: %s", hipGetErrorString(e));
    for (int64_t t = 0; t < T; ++t)
        for (int64_t r = 0; r < rows; ++r) c->h_Tz[(size_t)(t * rows + r)] = c->h_rowsum[(size_t)r] - tnz[(size_t)(t * rows + r)];
    c->compact_sorted = sort_by_count;
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
    c->sparse_at_upload = c->sparse;
    // dense data (more than a quarter of the bins hold events): dense forms only, unless the caller forces the lists.
    // Mostly empty data get the lists and the compacted templates whatever `sparse` says: with sparse = 0 they are
    // used by split scans only (k_scan_valid), every other path then visits every bin
    if (run > T * B / 4 && c->sparse != 2) { cleanup(); return BI_OK; }
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

}  // namespace
