// bi_single.h -- the synchronous single-point fast path behind bi_eval(P = 1).
#pragma once

namespace {


// The fused form: descriptors in the kernel arguments, reduction finished inside the launch, result written
// to pinned host memory -- one launch and one stream sync per call.
// Wait for the single-point launch with sequence number `seq` and fetch {ll, status} from the pinned block.
// The finishing thread stores them and then, with a system-scope release, the sequence number: poll that word
// instead of paying the runtime's stream-synchronise latency.  Falls back to the stream sync (which also reports a
// faulted kernel) if the word does not arrive in time; every 256th call synchronises anyway, so the runtime retires
// its completed commands at a steady pace.
int single_wait(bi_ctx* c, unsigned long long seq, double* out, int32_t* status) {
    char* res = (char*)c->slot_host;
    bool arrived = false;
    if (c->poll_result && !c->profiling) {
        const volatile unsigned long long* done = (const volatile unsigned long long*)(res + 16);
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
        for (unsigned spin = 0; !(arrived = (*done == seq)); ++spin) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() > t_end) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!arrived || (seq & 255ull) == 0) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (*(const volatile unsigned long long*)(res + 16) != seq)
        // the stream drained but nothing was published: report it, rather than hand back the zeroed block as a likelihood of 0
        return fail(c, BI_ERR_HIP, "single-point launch %llu finished without publishing its result", seq);
    *out = *(double*)res;
    if (status) *status = *(int32_t*)(res + 8);
    if (*(int32_t*)(res + 8) & BI_ST_INTERNAL) reset_mail(c);
    return BI_OK;
}

// wait = false (bi_eval_begin): return right after the launch; bi_eval_end collects the result with single_wait.
// numpy's pairwise_sum (loops_utils.h.src) on a host array: the partial last chunk of k_bb_chunk_sums
double pairwise_sum_host(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum_host(a, n2) + pairwise_sum_host(a + n2, n - n2);
}

// Can some bin have U_b == 0 (the other sources expecting exactly nothing) at this point?  Not if some other source has a
// positive rate and strictly positive templates at every corner that carries weight.
bool bb_zero_u_possible(const bi_ctx* c, int64_t cell_anchor, const double* w, const double* rates) {
    const int nc = 1 << (int)c->eff_axes.size();
    for (int s = 0; s < c->S; ++s) {
        if (s == c->bb_source || !(rates[s] > 0.0)) continue;
        bool positive = true;
        for (int corner = 0; corner < nc && positive; ++corner) {
            if (!(w[corner] > 0.0)) continue;                                 // a corner without weight adds exactly 0
            const int64_t a = cell_anchor + corner_offset(c, corner);
            positive = c->h_rowmin[(size_t)(a * c->S + s)] > 0.0;
        }
        if (positive) return false;
    }
    return true;
}

bool bb_zero_u_possible(const bi_ctx* c, const PointGeom& g, const double* rates) {
    return bb_zero_u_possible(c, g.cell_anchor, g.w.data(), rates);
}

// N(z) = sum_b a_b(z) exactly as the reference's `n_model_events[source_i].sum()` computes it (likelihood.py:645), for
// n points at once: one extra pass over the 2^d corner rows of the Monte-Carlo counts per point (k_bb_chunk_sums; points of
// one cell share them out of L2), a few KB per point back, a short host loop.  cell_anchor [n], w [n][nc] -> N [n].
int bb_exact_totals(bi_ctx* c, int64_t n, const int64_t* cell_anchor, const double* w, double* N) {
    if (n <= 0) return BI_OK;
    const int nc = 1 << (int)c->eff_axes.size();
    const int64_t B = c->B, n_full = B / kSumChunk, tail_n = B % kSumChunk;
    const int64_t n_blocks = n_full + (tail_n ? 1 : 0);
    if (n_blocks == 0) { std::fill(N, N + n, 0.0); return BI_OK; }
    if (n_blocks > 65535) return fail(c, BI_ERR_INVALID, "Beeston-Barlow total: %lld bins are more than one launch covers", (long long)B);
    std::vector<int64_t> rowoff((size_t)n * nc);
    for (int64_t q = 0; q < n; ++q)
        for (int corner = 0; corner < nc; ++corner) rowoff[(size_t)(q * nc + corner)] = (cell_anchor[q] + corner_offset(c, corner)) * c->Bp;
    const size_t per_point = (size_t)(n_full + tail_n);
    DevBuf d_in, d_out;
    int rc;
    if ((rc = dev_alloc(c, d_in, (size_t)n * nc * 16)) || (rc = dev_alloc(c, d_out, (size_t)n * per_point * sizeof(double)))) {
        dev_free(d_in); dev_free(d_out);
        return rc;
    }
    std::vector<double> h_out((size_t)n * per_point);
    hipError_t e = hipMemcpyAsync(d_in.p, rowoff.data(), (size_t)n * nc * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((char*)d_in.p + (size_t)n * nc * 8, w, (size_t)n * nc * 8, hipMemcpyHostToDevice, c->stream);
    double* sums = (double*)d_out.p;
    double* tails = sums + (size_t)n * n_full;
    for (int64_t q0 = 0; e == hipSuccess && q0 < n; q0 += 1 << 20) {          // (gridDim.x is generous; keep launches finite)
        const int64_t nq = std::min<int64_t>((int64_t)1 << 20, n - q0);
        hipLaunchKernelGGL(k_bb_chunk_sums, dim3((unsigned)nq, (unsigned)n_blocks), dim3(kThreads), 0, c->stream, (const double*)c->nm.p,
                           (const int64_t*)d_in.p + q0 * nc, (const double*)((char*)d_in.p + (size_t)n * nc * 8) + q0 * nc, nc, B,
                           sums + q0 * n_full, tails + q0 * tail_n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_out.data(), d_out.p, h_out.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_in); dev_free(d_out);
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "Beeston-Barlow total: %s", hipGetErrorString(e));
    for (int64_t q = 0; q < n; ++q) {
        const double* cs = h_out.data() + (size_t)q * n_full;
        bool have = false;
        double total = 0.0;
        for (int64_t k = 0; k < n_full; ++k) { total = have ? total + cs[k] : cs[k]; have = true; }
        if (tail_n) {
            const double t = pairwise_sum_host(h_out.data() + (size_t)n * n_full + (size_t)q * tail_n, tail_n);
            total = have ? total + t : t;
        }
        N[q] = total;
    }
    c->n_bb_exact += n;
    return BI_OK;
}

int bb_exact_total(bi_ctx* c, const PointGeom& g, double* N) {
    return bb_exact_totals(c, 1, &g.cell_anchor, g.w.data(), N);
}

// A point with an INFINITE rate (legal only beside a source that may go negative, likelihood.py:403-415).  The kernels fold
// rates into corner coefficients, (w_c r_s) p_c,b, which turns exact zeros of single corner templates into nan; the
// reference scales the INTERPOLATED template, r_s (sum_c w_c p_c,b).  So such points are answered here, the reference's
// way: the rows of the infinite-rate sources are interpolated exactly as scipy does (k_morph_store), and since an
// infinite expectation leaves only two possible results the rest is a classification per bin -- poisson.logpmf(n | mu):
// nan where mu is nan or negative, -inf where n is not a count, and for mu = +inf: -inf at n = 0, nan (inf - inf) at n > 0.
// -> *out = nan or -inf.  Plain binned likelihoods with finite templates only (the caller checks).
int inf_rate_value(bi_ctx* c, const PointGeom& g, const double* rates, int64_t ds, double* out) {
    const int nc = (int)g.w.size();
    std::vector<int> srcs;
    for (int s = 0; s < c->S; ++s)
        if (std::isinf(rates[s])) srcs.push_back(s);
    const int R = (int)srcs.size();
    const int64_t B = c->B;
    std::vector<int64_t> rowoff((size_t)R * nc);
    for (int r = 0; r < R; ++r)
        for (int corner = 0; corner < nc; ++corner)
            rowoff[(size_t)r * nc + corner] = ((g.cell_anchor + corner_offset(c, corner)) * c->S + srcs[(size_t)r]) * c->Bp;
    DevBuf d_row, d_w, d_out;
    auto cleanup = [&]() { dev_free(d_row); dev_free(d_w); dev_free(d_out); };
    int rc;
    if ((rc = dev_upload(c, d_row, rowoff)) || (rc = dev_upload(c, d_w, g.w)) ||
        (rc = dev_alloc(c, d_out, (size_t)R * std::max<int64_t>(B, 1) * sizeof(double)))) { cleanup(); return rc; }
    std::vector<double> rows((size_t)R * B), n((size_t)B);
    hipError_t e = hipSuccess;
    if (B > 0) {
        hipLaunchKernelGGL(k_morph_store, dim3((unsigned)((B + kThreads - 1) / kThreads), (unsigned)R), dim3(kThreads), 0, c->stream,
                           (const double*)c->ps.p, (const int64_t*)d_row.p, (const double*)d_w.p, nc, B, (double*)d_out.p);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(rows.data(), d_out.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(n.data(), (const double*)c->counts.p + ds * c->Bp, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "infinite-rate evaluation: %s", hipGetErrorString(e));
    bool any_nan = false;
    for (int64_t b = 0; b < B && !any_nan; ++b) {
        double mu = 0.0;                                        // the finite sources add finite numbers: they cannot change the class
        for (int r = 0; r < R; ++r) mu += rows[(size_t)r * B + b] * rates[srcs[(size_t)r]];
        const double nb = n[(size_t)b];
        if (!(mu >= 0.0) || nb != nb) any_nan = true;                     // scipy: mu invalid or n nan -> nan
        else if (nb < 0.0 || nb != std::floor(nb)) continue;              // outside the support: -inf
        else if (nb > 0.0) any_nan = true;                                // xlogy(n, inf) - inf
    }
    *out = any_nan ? std::numeric_limits<double>::quiet_NaN() : (B > 0 ? -std::numeric_limits<double>::infinity() : 0.0);
    return BI_OK;
}

inline bool has_infinite_rate(const double* rates, int S) {
    for (int s = 0; s < S; ++s)
        if (std::isinf(rates[s])) return true;
    return false;
}

using bi_clock = std::chrono::steady_clock;
inline int64_t ns_between(bi_clock::time_point a, bi_clock::time_point b) {
    return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
}

int eval_single_fused(bi_ctx* c, const PointGeom& g, const double* rates, int64_t ds, bool sparse, double* out, int32_t* status,
                      bool wait = true, bi_clock::time_point t_entry = bi_clock::now()) {
    const int S = c->S;
    const bool bb = c->bb_source >= 0;
    const int nc = (int)g.w.size();
    const int n0 = bb ? nc * (S - 1) : nc * S, n1 = bb ? nc : 0, n2 = bb ? nc : 0;
    const int64_t row_stride = sparse ? c->h_c_np[(size_t)ds] : c->Bp;
    const int64_t row_base = sparse ? c->h_c_off[(size_t)ds] : 0;
    const int tiles = (int)(row_stride / kTile);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    // launch shape of ONE pass over the rows: every block gets the same number of tiles and all blocks are resident at
    // once (a second, thin round of blocks is a tail the size of a block's lifetime).  Measured on C2 (1954 tiles):
    // 2 tiles per block on ~4 blocks per CU takes 48 us from launch to result, 1 tile per block on 7.6 per CU 54 us.
    const int64_t target = (int64_t)c->prop.multiProcessorCount * c->single_blocks_per_cu;
    const int64_t tiles_per_block = std::max<int64_t>(1, (tiles + target - 1) / target);
    int64_t want = (tiles + tiles_per_block - 1) / tiles_per_block;
    if (want >= 8) want = (want + 7) / 8 * 8;      // a block's tiles then all lie in one of the 8 row regions (block b -> XCD b % 8):
                                                   // 984 blocks take 48 us where 977 take 59
    const int nbx = (int)std::min<int64_t>(std::min<int64_t>(want, tiles), slots);
    int rc;
    if (!c->slot_host) {
        HIP_TRY(c, hipHostMalloc(&c->slot_host, 4096, hipHostMallocDefault));
        c->slot_host_bytes = 4096;
    }
    const bool fuse = nbx <= c->fuse_max_blocks && nbx <= kMailSlots && !ensure_mail(c);
    if (!fuse && ((rc = dev_alloc(c, c->slot_partial, (size_t)slots * sizeof(double))) ||
                  (rc = dev_alloc(c, c->slot_pflags, (size_t)slots * sizeof(unsigned)))))
        return rc;
    SingleDesc d;
    const int64_t n_rows = c->A * S;
    int k = 0;
    double zsum = 0.0;
    for (int corner = 0; corner < nc; ++corner) {
        const int64_t a = g.cell_anchor + corner_offset(c, corner);
        for (int s = 0; s < S; ++s) {
            if (bb && s == c->bb_source) continue;
            d.rowoff[k] = row_base + (a * S + s) * row_stride;
            d.coef[k] = g.w[(size_t)corner] * rates[s];
            if (sparse) zsum += d.coef[k] * c->h_Tz[(size_t)(ds * n_rows + a * S + s)];
            ++k;
        }
    }
    d.aux[0] = d.aux[1] = 1.0;
    if (bb) {
        double Ntot = 0.0;
        for (int corner = 0; corner < nc; ++corner) {
            const int64_t a = g.cell_anchor + corner_offset(c, corner);
            d.rowoff[n0 + corner] = (a * S + c->bb_source) * c->Bp;
            d.coef[n0 + corner] = g.w[(size_t)corner];
            d.rowoff[n0 + n1 + corner] = a * c->Bp;
            d.coef[n0 + n1 + corner] = g.w[(size_t)corner];
            const double term = c->h_nm_tot[(size_t)a] * g.w[(size_t)corner];
            Ntot = Ntot + term;
        }
        if (c->bb_exact == 1 || (c->bb_exact == 2 && bb_zero_u_possible(c, g, rates))) {
            if ((rc = bb_exact_total(c, g, &Ntot))) return rc;
        }
        d.aux[0] = rates[c->bb_source] / Ntot;
        d.aux[1] = Ntot;
    }
    d.slot_lg = c->h_lgsum[(size_t)ds] + zsum;
    if (c->unbinned) {
        double rsum = 0.0;
        for (int s = 0; s < S; ++s) rsum += rates[s];
        d.slot_lg = rsum;
    }
    char* res = (char*)c->slot_host;
    *(double*)res = 0.0;
    *(int64_t*)(res + 8) = 0;
    *(unsigned long long*)(res + 16) = 0ull;
    d.flags = fuse ? (unsigned*)c->mail_flags.p : nullptr;
    d.out = (double*)res;
    d.status = (int32_t*)(res + 8);
    d.done = (unsigned long long*)(res + 16);
    d.seq = ++c->slot_seq;
    LaunchArgs a{};
    a.ps = sparse ? (const double*)c->ps_c.p : (const double*)c->ps.p;
    a.nm = (const double*)c->nm.p;
    a.counts = (sparse ? (const double*)c->cnt_c.p + c->h_cnt_off[(size_t)ds] : (const double*)c->counts.p + ds * c->Bp);
    a.partial = fuse ? (double*)c->mail.p : (double*)c->slot_partial.p;
    a.pflags = (unsigned*)c->slot_pflags.p;
    a.B = c->B; a.Bp = c->Bp; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.n_tiles = tiles; a.chunks = (int)c->tile_chunks;
    a.outlier = c->outlier;
    a.nan_S = (c->unbinned && !c->ps_finite) ? c->S : 0;
    if (fuse) arm_mail(c, a);
    const bool nt = !sparse && c->nt_loads != 0;
    // Repeated evaluations in one grid cell (a minimizer's access pattern): let most of the cell's rows keep the default
    // cache policy so that they stay in the 256 MiB Infinity Cache between calls (the rest, and every call into a new
    // cell, stream with the nontemporal hint).  Measured on C2: kernel 50.9 -> 45.1 us with 28 of 32 rows kept.
    a.n_keep = 0;
    if (nt && !bb && c->keep_rows != 0) {
        const bool same_cell = c->last_single_cell == g.cell_anchor && c->last_single_ds == ds;
        const int64_t fit = (int64_t)(0.85 * 256.0 * 1048576.0 / ((double)row_stride * sizeof(double)));
        if (same_cell) a.n_keep = (int)std::min<int64_t>(c->keep_rows > 0 ? c->keep_rows : fit, n0);
    }
    c->last_single_cell = g.cell_anchor;
    c->last_single_ds = ds;
    const dim3 grid((unsigned)nbx), block(kThreads);
    const auto t_launch = bi_clock::now();
    {
        EventScope ev(c);
        launch_morph_single(c, bb, nt, fuse, grid, a, d);
    }
    if (!fuse)
        hipLaunchKernelGGL(k_finish_single, dim3(1), block, 0, c->stream, (const double*)a.partial, (const unsigned*)a.pflags,
                           nbx, d.slot_lg, d.out, d.status, d.done, d.seq);
    HIP_TRY(c, hipGetLastError());
    const auto t_wait = bi_clock::now();
    c->single_ns[0] += ns_between(t_entry, t_launch);
    c->single_ns[1] += ns_between(t_launch, t_wait);
    ++c->single_calls;
    if (!wait) {
        c->pending = 1;
        c->pending_seq = d.seq;
        return BI_OK;
    }
    rc = single_wait(c, d.seq, out, status);
    c->single_ns[2] += ns_between(t_wait, bi_clock::now());
    return rc;
}

// One point, synchronous: the call shape of `lf(**kwargs)` inside a minimizer (inference.py:111-122 makes
// ~500 of them per fit).  Same kernels as the batched path, but the descriptors live in a persistent
// device slot fed from pinned memory: one small H2D, two launches, one 16-byte D2H, one sync.
// wait = false: launch only (bi_eval_begin).  Answers that need no launch -- and the rare fallback path, which then
// runs synchronously -- are parked in the context (pending = 2) for bi_eval_end.
int eval_single(bi_ctx* c, const double* z, const double* rate_scale, int64_t ds, double* out, int32_t* status, bool wait = true) {
    const auto t_entry = bi_clock::now();
    double parked_ll = 0.0;
    int32_t parked_st = 0;
    if (!wait) { out = &parked_ll; status = &parked_st; c->pending = 2; }
    struct Park {   // on every return path of a launch-less answer: keep it for bi_eval_end
        bi_ctx* c; const bool on; double* ll; int32_t* st;
        ~Park() { if (on && c->pending == 2) { c->pending_ll = *ll; c->pending_status = *st; } }
    } park{c, !wait, &parked_ll, &parked_st};
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
    if (has_infinite_rate(rates, S) && c->bb_source < 0 && !c->unbinned && c->ps_finite && c->dense_counts)
        return inf_rate_value(c, g, rates, ds, out);

    const bool bb = c->bb_source >= 0;
    const int nc = (int)g.w.size();
    const int n0 = bb ? nc * (S - 1) : nc * S, n1 = bb ? nc : 0, n2 = bb ? nc : 0, NS = n0 + n1 + n2;
    bool any_neg = false;
    for (int q = 0; q < S; ++q) any_neg |= (c->allow_neg[(size_t)q] != 0);
    const bool sparse = c->sparse && c->compact_ready && c->ps_nonneg && !bb && !any_neg && !c->unbinned;
    if (!sparse && !c->dense_counts) return fail(c, BI_ERR_STATE, "dataset counts are not resident in dense form");
    const int64_t row_stride = sparse ? c->h_c_np[(size_t)ds] : c->Bp;
    const int64_t row_base = sparse ? c->h_c_off[(size_t)ds] : 0;
    const int tiles = (int)(row_stride / kTile);
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nbx = (int)std::min<int64_t>(tiles, slots);

    if (NS <= kMaxSingleStreams && c->single_kernel) return eval_single_fused(c, g, rates, ds, sparse, out, status, wait, t_entry);

    // general fallback (more streams than fit the kernel-argument block):
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
        if (c->bb_exact == 1 || (c->bb_exact == 2 && bb_zero_u_possible(c, g, rates))) {
            if ((rc = bb_exact_total(c, g, &Ntot))) return rc;
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
    a.B = c->B; a.Bp = c->Bp; a.n0 = n0; a.n1 = n1; a.n2 = n2; a.n_tiles = tiles; a.chunks = (int)c->tile_chunks;
    a.outlier = c->outlier;
    a.nan_S = (c->unbinned && !c->ps_finite) ? c->S : 0;
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
