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

#include "bi_common.h"
#include "bi_prim.h"
#include "bi_k_misc.h"
#include "bi_geometry.h"
#include "bi_launch.h"
#include "bi_sparse.h"
#include "bi_single.h"
#include "bi_planning.h"
#include "bi_planning_device.h"
#include "bi_grad_mfma.h"
#include "bi_grad_bb.h"
#include "bi_params.h"

namespace {

// dense counts [T][Bp] are resident: per-dataset sum lgamma(n+1), then the sparse forms
int finish_counts(bi_ctx* c, int64_t T) {
    int rc;
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
    dev_free(c->scratch); dev_free(c->scratch2); dev_free(c->logmu); dev_free(c->toy_blocks_done); dev_free(c->ev_perm);
    dev_free(c->slot_dev); dev_free(c->slot_partial); dev_free(c->slot_pflags); dev_free(c->slot_counter); dev_free(c->space_edges);
    dev_free(c->mail); dev_free(c->mail_flags);
    dev_free(c->ps_sorted); dev_free(c->cnt_sorted);
    dev_free(c->sim_coords); dev_free(c->sim_source);
    if (c->slot_host) (void)hipHostFree(c->slot_host);
    if (c->pack_host) (void)hipHostFree(c->pack_host);
    if (c->bounce_host) (void)hipHostFree(c->bounce_host);
    if (c->plan_host) (void)hipHostFree(c->plan_host);
    dev_free(c->pack_dev);
    dev_free(c->nz_idx); dev_free(c->nz_n); dev_free(c->nz_off); dev_free(c->ps_c); dev_free(c->cnt_c); dev_free(c->tm_entries); dev_free(c->tm_off); dev_free(c->tmm_entries); dev_free(c->tmm_off);
    dev_free(c->pt_grid); dev_free(c->pt_mus); dev_free(c->pt_coff); dev_free(c->pt_allow); dev_free(c->pt_c_off);
    dev_free(c->pt_cnt_off); dev_free(c->pt_c_np); dev_free(c->pt_Tz); dev_free(c->pt_rowsum); dev_free(c->pt_rowmin); dev_free(c->pt_nm_tot);
    for (void* q : c->user_allocs) (void)hipFree(q);   // bi_device_alloc buffers nobody freed
    c->user_allocs.clear();
    for (auto& q : c->cache) (void)hipFree(q.p);  // last: the dev_free calls above may have parked buffers
    c->cache.clear();
    for (auto& ev : c->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (hipEvent_t ev : c->tp_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
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
    const ParamDef* p = find_param(name);
    if (!p) return fail(c, BI_ERR_INVALID, "unknown parameter %s", name);
    if (!p->set) return fail(c, BI_ERR_INVALID, "parameter %s is read-only", name);
    return p->set(c, v);
}

int64_t bi_get_param(bi_ctx* c, const char* name) {
    if (!c || !name) return INT64_MIN;
    const ParamDef* p = find_param(name);
    if (!p || !p->get) {
        fail(c, BI_ERR_INVALID, p ? "parameter %s is write-only" : "unknown parameter %s", name);
        return INT64_MIN;
    }
    return p->get(c);
}

int bi_list_params(char* buf, int len) {
    std::string all;
    for (const ParamDef& p : kParams) {
        all += p.name;
        all += p.access == kParamRW ? " rw\n" : (p.access == kParamRead ? " r\n" : " w\n");
    }
    if (buf && len > 0) {
        const size_t n = std::min<size_t>(all.size(), (size_t)len - 1);
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int)all.size() + 1;
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
    c->ev_sorted = false;
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
        c->h_rowmin.assign((size_t)rows, 0.0);
        c->ps_nonneg = true;
        c->ps_finite = true;
        for (int64_t r = 0; r < rows; ++r) {
            c->h_rowsum[(size_t)r] = st[(size_t)r * 3];
            c->h_rowmin[(size_t)r] = st[(size_t)r * 3 + 1];
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

int bi_set_bb_totals(bi_ctx* c, const double* totals) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (c->bb_source < 0) return fail(c, BI_ERR_INVALID, "the model has no Beeston-Barlow source");
    if (!totals) return fail(c, BI_ERR_INVALID, "totals is NULL");
    c->h_nm_tot.assign(totals, totals + c->A);
    ++c->epoch;
    return BI_OK;
}

int bi_get_bb_totals(bi_ctx* c, double* totals) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (c->bb_source < 0 || !totals) return fail(c, BI_ERR_INVALID, "no Beeston-Barlow source / totals is NULL");
    std::copy(c->h_nm_tot.begin(), c->h_nm_tot.end(), totals);
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
    return finish_counts(c, T);
}


// ---- set_data on the device ---------------------------------------------------------------------

int bi_set_analysis_space(bi_ctx* c, int k, const int32_t* n_edges, const double* edges) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (k < 1 || k > kMaxDim || !n_edges || !edges) return fail(c, BI_ERR_INVALID, "need 1..%d axes with edges", kMaxDim);
    int64_t bins = 1;
    std::vector<double> flat;
    std::vector<int32_t> ne(n_edges, n_edges + k);
    for (int i = 0; i < k; ++i) {
        if (ne[(size_t)i] < 2) return fail(c, BI_ERR_INVALID, "axis %d needs at least two edges", i);
        bins *= ne[(size_t)i] - 1;
    }
    const double* e = edges;
    for (int i = 0; i < k; ++i) {
        for (int j = 1; j < ne[(size_t)i]; ++j)
            if (!(e[j] > e[j - 1])) return fail(c, BI_ERR_INVALID, "bin edges of axis %d are not strictly ascending", i);
        flat.insert(flat.end(), e, e + ne[(size_t)i]);
        e += ne[(size_t)i];
    }
    if (bins != c->B) return fail(c, BI_ERR_INVALID, "analysis space has %lld bins, the model %lld", (long long)bins, (long long)c->B);
    HIP_TRY(c, hipSetDevice(c->device));
    c->space_k = k;
    c->space_n_edges = ne;
    if ((rc = dev_upload(c, c->space_edges, flat))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BI_OK;
}

int bi_upload_events(bi_ctx* c, int64_t N, const double* coords) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (c->space_k < 1) return fail(c, BI_ERR_STATE, "bi_set_analysis_space first");
    if (c->unbinned) return fail(c, BI_ERR_STATE, "the context holds an unbinned likelihood");
    if (N < 0 || (N > 0 && !coords)) return fail(c, BI_ERR_INVALID, "bad N / coords");
    HIP_TRY(c, hipSetDevice(c->device));
    c->data_ready = false;
    ++c->epoch;
    const size_t bytes = (size_t)c->Bp * sizeof(double);
    if ((rc = dev_alloc(c, c->counts, bytes))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->counts.p, 0, bytes, c->stream));
    if (N > 0) {
        DevBuf d_ev;
        if ((rc = dev_alloc(c, d_ev, (size_t)N * c->space_k * sizeof(double)))) return rc;
        hipError_t e = hipMemcpyAsync(d_ev.p, coords, (size_t)N * c->space_k * sizeof(double), hipMemcpyHostToDevice, c->stream);
        HistArgs h{};
        h.k = c->space_k;
        int off = 0;
        for (int i = 0; i < c->space_k; ++i) { h.n_edges[i] = c->space_n_edges[(size_t)i]; h.edge_off[i] = off; off += h.n_edges[i]; }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_histogram, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                               (const double*)d_ev.p, N, h, (const double*)c->space_edges.p, (double*)c->counts.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // coords are borrowed for the call only
        dev_free(d_ev);
        if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_upload_events: %s", hipGetErrorString(e));
    }
    return finish_counts(c, 1);
}

int bi_histogram_events(bi_ctx* c, int k, const int32_t* n_edges, const double* edges, int64_t N, const double* coords,
                        double* counts) {
    if (!c) return BI_ERR_INVALID;
    if (c->pending) return fail(c, BI_ERR_STATE, "a bi_eval_begin is outstanding on this context: call bi_eval_end first");
    if (k < 1 || k > kMaxDim || !n_edges || !edges || !counts) return fail(c, BI_ERR_INVALID, "need 1..%d axes with edges, and a counts buffer", kMaxDim);
    if (N < 0 || (N > 0 && !coords)) return fail(c, BI_ERR_INVALID, "bad N / coords");
    HistArgs h{};
    h.k = k;
    int64_t bins = 1;
    int off = 0;
    for (int i = 0; i < k; ++i) {
        if (n_edges[i] < 2) return fail(c, BI_ERR_INVALID, "axis %d needs at least two edges", i);
        for (int j = 1; j < n_edges[i]; ++j)
            if (!(edges[off + j] > edges[off + j - 1])) return fail(c, BI_ERR_INVALID, "bin edges of axis %d are not strictly ascending", i);
        h.n_edges[i] = n_edges[i];
        h.edge_off[i] = off;
        off += n_edges[i];
        bins *= n_edges[i] - 1;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf d_ev, d_edges, d_counts;
    int rc;
    if ((rc = dev_alloc(c, d_counts, (size_t)bins * sizeof(double))) || (rc = dev_alloc(c, d_edges, (size_t)off * sizeof(double))) ||
        (N > 0 && (rc = dev_alloc(c, d_ev, (size_t)N * k * sizeof(double))))) {
        dev_free(d_counts); dev_free(d_edges); dev_free(d_ev);
        return rc;
    }
    hipError_t e = hipMemsetAsync(d_counts.p, 0, (size_t)bins * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_edges.p, edges, (size_t)off * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && N > 0) {
        e = hipMemcpyAsync(d_ev.p, coords, (size_t)N * k * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_histogram, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                               (const double*)d_ev.p, N, h, (const double*)d_edges.p, (double*)d_counts.p);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_counts.p, (size_t)bins * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    dev_free(d_counts); dev_free(d_edges); dev_free(d_ev);
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_histogram_events: %s", hipGetErrorString(e));
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
    return plan_points(c, P, z, rate_scale, dataset, out);
}

int bi_plan_points_share(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, int share_rank,
                         int share_world, bi_plan** out) {
    if (share_world < 1) return fail(c, BI_ERR_INVALID, "share_world must be >= 1");
    return plan_points(c, P, z, rate_scale, dataset, out, /*transient=*/false, share_rank, share_world);
}

int bi_plan_points_resident(bi_ctx* c, int64_t P, const double* z_dev, const double* rate_scale_dev, const int64_t* dataset_dev,
                            int share_rank, int share_world, bi_plan** out) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!out) return fail(c, BI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (P < 0 || P > (int64_t)1 << 30) return fail(c, BI_ERR_INVALID, "P outside [0, 2^30]");
    if (share_world < 1 || share_rank < 0 || share_rank >= share_world)
        return fail(c, BI_ERR_INVALID, "share %d outside [0,%d)", share_rank, share_world);
    if (c->d > 0 && P > 0 && !z_dev) return fail(c, BI_ERR_INVALID, "z_dev is NULL");
    bool any_neg = false;
    for (int q = 0; q < c->S; ++q) any_neg |= (c->allow_neg[(size_t)q] != 0);
    HIP_TRY(c, hipSetDevice(c->device));
    if (P == 0) return plan_points(c, 0, nullptr, nullptr, nullptr, out);
    const void* ptrs[3] = {c->d > 0 ? (const void*)z_dev : nullptr, rate_scale_dev, dataset_dev};
    const char* names[3] = {"z_dev", "rate_scale_dev", "dataset_dev"};
    for (int i = 0; i < 3; ++i) {
        if (!ptrs[i]) continue;
        hipPointerAttribute_t at;
        const hipError_t e = hipPointerGetAttributes(&at, ptrs[i]);
        if (e != hipSuccess || (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) || at.device != c->device) {
            (void)hipGetLastError();
            return fail(c, BI_ERR_INVALID, "%s is not device memory of GPU %d", names[i], c->device);
        }
    }
    const bool sparse = c->sparse && c->compact_ready && c->ps_nonneg && !c->unbinned && c->bb_source < 0 && !any_neg;
    if (!sparse && !c->dense_counts)
        return fail(c, BI_ERR_STATE, "the datasets exist only as non-empty-bin lists (device-generated toys): point "
                                     "evaluations need the compacted templates (sparse mode, budget) or bi_eval_datasets");
    // (sources that may go negative: an infinite rate among the points is found by the planner itself and refused;
    //  Beeston-Barlow: refused when some point needs the host planner's exact totals)
    rc = plan_points_device(c, P, z_dev, rate_scale_dev, dataset_dev, sparse, out, share_rank, share_world, true, false);
    if (rc == kPlanNeedsHost)
        return fail(c, BI_ERR_INVALID, "resident points are planned on the device: this Beeston-Barlow batch has points at which some bin can "
                                       "have U_b == 0 (or bb_exact = 1) and needs the host planner's exact totals (bi_plan_points)");
    return rc;
}

int bi_plan_share_info(const bi_plan* p, int64_t* n_valid, int64_t* lo, int64_t* hi) {
    if (!p || !p->shared) return BI_ERR_INVALID;
    if (n_valid) *n_valid = p->n_valid;
    if (lo) *lo = p->share_lo;
    if (hi) *hi = p->share_hi;
    return BI_OK;
}

int bi_plan_unsort(bi_ctx* c, bi_plan* plan, const double* gathered_dev, int64_t stride, double* full_dev) {
    if (!c || !plan) return BI_ERR_INVALID;
    if (!plan->shared) return fail(c, BI_ERR_INVALID, "bi_plan_unsort: not a share of a dealt scan (bi_plan_points_share)");
    const int64_t per_rank = (plan->n_valid + plan->share_world - 1) / plan->share_world;
    if (!gathered_dev || !full_dev || stride < per_rank) return fail(c, BI_ERR_INVALID, "bi_plan_unsort: NULL buffer or stride %lld < %lld", (long long)stride, (long long)per_rank);
    HIP_TRY(c, hipSetDevice(c->device));
    if (plan->P > 0)
        hipLaunchKernelGGL(k_unsort_share, dim3((unsigned)((plan->P + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, gathered_dev,
                           stride, plan->share_world, plan->n_valid, plan->P, (const int64_t*)plan->sorted_idx.p, full_dev);
    HIP_TRY(c, hipGetLastError());
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
    a.nan_S = (c->unbinned && !c->ps_finite) ? c->S : 0;
    a.n0 = bb ? nc * (c->S - 1) : nc * c->S;
    a.n1 = bb ? nc : 0; a.n2 = bb ? nc : 0;
    a.n_tiles = n_tiles_of(c);
    a.chunks = (int)c->tile_chunks;
    const int NS = a.n0 + a.n1 + a.n2;
    if (plan->use_scan) {
        bi_plan::Class& k = plan->classes[0];
        ScanArgs sa{};
        sa.ps = a.ps; sa.counts = a.counts;
        if (plan->sorted) {
            if (c->sorted_epoch != c->epoch || !c->sorted_ok) return fail(c, BI_ERR_STATE, "plan is stale: the count-sorted rows are gone");
            sa.ps = (const double*)c->ps_sorted.p; sa.counts = (const double*)c->cnt_sorted.p;
            ++c->n_sorted_scans;
        }
        sa.rowoff = (const int64_t*)k.rowoff.p; sa.coef = (const double*)k.coef.p;
        sa.item_cnt = (const int64_t*)k.item_cnt.p; sa.item_tiles = (const int32_t*)k.item_tiles.p;
        sa.grp_first = (const int64_t*)plan->grp_first.p; sa.grp_items = (const int32_t*)plan->grp_items.p;
        sa.partial = (double*)k.partial.p; sa.NS = NS; sa.nslots = k.nbx;
        HIP_TRY(c, hipMemsetAsync(k.partial.p, 0, (size_t)k.n_items * k.nbx * k.G * sizeof(double), c->stream));
        {
            EventScope ev(c);
            ++c->n_scan_launches;
            const dim3 sgrid((unsigned)(k.nbx / 4), (unsigned)plan->n_groups);
            if (plan->by_count) {
                // rows ordered by count (all bins of dense data, or the compacted non-empty bins): 64-bin strips, four items at a time
                sa.n_groups = (int)plan->n_groups;
                sa.xcd_mode = (int)c->scan_xcd;
                sa.share_slow = (int)c->scan_share_slow;
                const int64_t scan_blocks = (int64_t)(k.nbx / 4) * (c->scan_xcd == 2 ? (plan->n_groups + 7) / 8 * 8 : plan->n_groups);
                launch_scan_sorted(c, NS, dim3((unsigned)((scan_blocks + 7) / 8 * 8)), sa);
            } else
                launch_scan_mfma(c, plan->scan_cb == 2 ? 2 : 4, plan->sparse, NS, sgrid, sa);
        }
        hipLaunchKernelGGL(k_finish_scan, dim3((unsigned)((k.n_items + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0,
                           c->stream, (const double*)k.partial.p, k.nbx, k.n_items, (const int64_t*)k.perm.p,
                           (const double*)k.slot_lg.p, out);
    }
    if (plan->bb_kgt && !plan->classes.empty()) {
        // Beeston-Barlow batch on the matrix cores: one launch over (tile ranges, groups, quads of a group's items), then the
        // ordinary finish of the per-block partial sums and status bits
        bi_plan::Class& k = plan->classes[0];
        BbScanArgs ba{};
        ba.ps = a.ps; ba.nm = a.nm; ba.counts = a.counts;
        ba.rowoff = (const int64_t*)k.rowoff.p; ba.coef = (const double*)k.coef.p; ba.aux = (const double*)k.aux.p;
        ba.item_cnt = (const int64_t*)k.item_cnt.p;
        ba.grp_first = (const int64_t*)plan->grp_first.p; ba.grp_items = (const int32_t*)plan->grp_items.p;
        ba.partial = (double*)k.partial.p; ba.pflags = (unsigned*)k.pflags.p;
        ba.B = c->B; ba.n0 = a.n0; ba.nc = a.n1; ba.n_tiles = (int)((c->B + 15) / 16);
        const int64_t quads = std::max<int64_t>(1, (plan->max_group_items + 3) / 4);
        {
            EventScope ev(c);
            ++c->n_bb_scan_launches;
            launch_scan_bb(c, plan->bb_kgt, dim3((unsigned)k.nbx, (unsigned)plan->n_groups, (unsigned)quads), ba);
        }
        const int64_t n_slots = k.n_items * k.G;
        const int lanes = k.nbx > 64 ? kThreads : 64;
        const int per_block = kThreads / lanes;
        hipLaunchKernelGGL(k_finish, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(kThreads), 0, c->stream,
                           (const double*)k.partial.p, (const unsigned*)k.pflags.p, k.nbx, k.G, lanes, n_slots, (const int64_t*)k.perm.p,
                           (const double*)k.slot_lg.p, out, (int32_t*)plan->status.p);
    }
    for (auto& k : plan->classes) {
        if (plan->use_scan || plan->bb_kgt) break;
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
            // nontemporal row loads when no two items touch the same anchor model -- or when an item streams more than the
            // Infinity Cache can keep (> 1 GiB: a shared anchor's rows are gone before the other item asks for them;
            // C5's 5.65 GB passes: 6.4 instead of 6.2 TB/s with 4 ... 16 items per call)
            const bool huge_items = (int64_t)sizeof(double) * (NS + 1) * c->Bp > ((int64_t)1 << 30);
            const bool nt = !plan->sparse && (c->nt_loads == 1 || (c->nt_loads == 2 && (plan->no_reuse || huge_items)));
            // few items (a fit's or a bench step's batch): the last block of every item finishes it inside the launch
            // (at most 256 collecting blocks at a time: they wait for their siblings, and must never be able to hold
            // every slot of the chip while siblings still need one, whatever order the blocks are dispatched in)
            const bool fuse = c->fuse_finish && k.n_items <= 256 && k.n_items * k.nbx * k.G <= kMailSlots &&
                              k.n_items * k.G <= kMailFlagWords && !ensure_mail(c);
            if (fuse) {
                b.fin_mail = (double*)c->mail.p;
                b.fin_flags = (unsigned*)c->mail_flags.p;
                b.fin_perm = (const int64_t*)k.perm.p;
                b.fin_slot_lg = (const double*)k.slot_lg.p;
                b.fin_out = out;
                b.fin_status = (int32_t*)plan->status.p;
                arm_mail(c, b);
            }
            launch_morph_g(c, k.G, b, dim3((unsigned)k.nbx, (unsigned)ni), bb, nt);
            if (fuse) continue;
            const int64_t n_slots = ni * k.G;
            const int lanes = k.nbx > 64 ? kThreads : 64;
            const int per_block = kThreads / lanes;
            hipLaunchKernelGGL(k_finish, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(kThreads), 0,
                               c->stream, (const double*)b.partial, (const unsigned*)b.pflags, k.nbx, k.G, lanes, n_slots,
                               (const int64_t*)k.perm.p + i0 * k.G, (const double*)k.slot_lg.p + i0 * k.G, out,
                               (int32_t*)plan->status.p);
        }
    }
    if (plan->valid) {
        // split dense scan: the classes above were the non-empty-bin pass; now every bin of every cell, on the matrix cores
        bi_plan::Class& k = plan->classes[0];
        ValidArgs va{};
        va.ps = (const double*)c->ps.p;
        va.rowoff = (const int64_t*)k.rowoff_full.p; va.coef = (const double*)k.coef.p;
        va.grp_first = (const int64_t*)plan->grp_first.p; va.grp_items = (const int32_t*)plan->grp_items.p;
        va.bad = (unsigned*)plan->bad.p; va.NS = NS; va.nslots = plan->valid_nslots;
        va.n_strips = n_tiles_of(c) * (kTile / 64);
        HIP_TRY(c, hipMemsetAsync(plan->bad.p, 0, (size_t)k.n_items * k.G * sizeof(unsigned), c->stream));
        {
            EventScope ev(c);
            ++c->n_valid_launches;
            const dim3 vgrid((unsigned)(plan->valid_nslots / 4), (unsigned)plan->n_groups);
            launch_scan_valid(c, NS, vgrid, va);
        }
        const int64_t n_slots = k.n_items * k.G;
        hipLaunchKernelGGL(k_apply_bad, dim3((unsigned)((n_slots + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                           (const unsigned*)plan->bad.p, (const int64_t*)k.perm.p, n_slots, out);
    }
    if (plan->shared) {
        // (rejected points belong to no share: bi_plan_unsort answers them)
    } else if (plan->n_bad > 0 && plan->device_planned)
        hipLaunchKernelGGL(k_fill_bad_by_status, dim3((unsigned)((plan->P + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                           (const int32_t*)plan->status.p, plan->P, out);
    else if (plan->n_bad > 0)
        hipLaunchKernelGGL(k_fill_const, dim3((unsigned)((plan->n_bad + 255) / 256)), dim3(256), 0, c->stream, out,
                           (const int64_t*)plan->bad_idx.p, plan->n_bad, -std::numeric_limits<double>::infinity());
    if (plan->n_nan > 0)
        hipLaunchKernelGGL(k_fill_const, dim3((unsigned)((plan->n_nan + 255) / 256)), dim3(256), 0, c->stream, out,
                           (const int64_t*)plan->nan_idx.p, plan->n_nan, std::numeric_limits<double>::quiet_NaN());
    HIP_TRY(c, hipGetLastError());
    return BI_OK;
}

// BI_ST_INTERNAL has been reported for this plan: empty the mailbox, and take the bit out of the plan's (sticky) status words
static void internal_reported(bi_ctx* c, bi_plan* plan) {
    reset_mail(c);
    if (!plan->P) return;
    if (plan->host_results) {
        int32_t* st = (int32_t*)plan->status.p;
        for (int64_t p = 0; p < plan->P; ++p) st[p] &= ~BI_ST_INTERNAL;
    } else {
        hipLaunchKernelGGL(k_status_clear, dim3((unsigned)std::min<int64_t>(1024, (plan->P + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                           c->stream, (int32_t*)plan->status.p, plan->P, (int32_t)BI_ST_INTERNAL);
        (void)hipStreamSynchronize(c->stream);
    }
}

int bi_plan_read(bi_ctx* c, bi_plan* plan, double* out, int32_t* status) {
    if (!c || !plan) return BI_ERR_INVALID;
    if (out && plan->shared) return fail(c, BI_ERR_INVALID, "bi_plan_read: a share's results are in sorted order: gather them and call bi_plan_unsort");
    HIP_TRY(c, hipSetDevice(c->device));
    if (out && plan->P)
        HIP_TRY(c, hipMemcpyAsync(out, plan->out.p, (size_t)plan->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status && plan->P)
        HIP_TRY(c, hipMemcpyAsync(status, plan->status.p, (size_t)plan->P * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (status) {
        for (int64_t p = 0; p < plan->P; ++p)
            if (status[p] & BI_ST_INTERNAL) { internal_reported(c, plan); break; }
        return BI_OK;
    }
    int32_t any = 0;
    return bi_plan_status(c, plan, &any);
}

int bi_plan_status(bi_ctx* c, bi_plan* plan, int32_t* status_or) {
    if (!c || !plan) return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int32_t any = 0;
    if (plan->P) {
        if (plan->host_results) {              // the status words already sit in pinned host memory
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const int32_t* st = (const int32_t*)plan->status.p;
            for (int64_t p = 0; p < plan->P; ++p) any |= st[p];
        } else {
            DevBuf d_or;
            int rc = dev_alloc(c, d_or, 64);
            if (rc) return rc;
            hipError_t e = hipMemsetAsync(d_or.p, 0, 4, c->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_status_or, dim3((unsigned)std::min<int64_t>(1024, (plan->P + kThreads - 1) / kThreads)), dim3(kThreads),
                                   0, c->stream, (const int32_t*)plan->status.p, plan->P, (int32_t*)d_or.p);
                e = hipGetLastError();
            }
            // (the word comes back through the planner's pinned report block: no copy into pageable memory, no stream synchronisation)
            if (e == hipSuccess) e = plan_report(c, ReportPiece{(const uint32_t*)d_or.p, 1, 0});
            if (e == hipSuccess) any = *(const int32_t*)c->plan_host;
            dev_free(d_or);
            if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_plan_status: %s", hipGetErrorString(e));
        }
    }
    if (any & BI_ST_INTERNAL) internal_reported(c, plan);
    if (status_or) *status_or = any;
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
    int rc = plan_points(c, P, z, rate_scale, dataset, &plan, /*transient=*/true);
    if (rc) return rc;
    rc = bi_run_plan(c, plan, nullptr);
    if (!rc && plan->host_results) {   // small batch: the finish kernel wrote out / status into pinned host memory
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, BI_ERR_HIP, "bi_eval: %s", hipGetErrorString(e));
        if (!rc && P) {
            memcpy(out, plan->out.p, (size_t)P * sizeof(double));
            if (status) memcpy(status, plan->status.p, (size_t)P * sizeof(int32_t));
            const int32_t* st = (const int32_t*)plan->status.p;
            for (int64_t p = 0; p < P; ++p)
                if (st[p] & BI_ST_INTERNAL) { reset_mail(c); break; }
        }
    } else if (!rc) {
        rc = bi_plan_read(c, plan, out, status);
    }
    bi_plan_destroy(c, plan);
    return rc;
}


// ---- value + analytic gradient in one pass ----------------------------------------------------

// The two halves of bi_eval(P = 1), for callers that evaluate several contexts at once (a sum of likelihoods, one
// context per term): begin on every context, then end on every context -- the launches overlap.
int bi_eval_begin(bi_ctx* c, const double* z, const double* rate_scale, int64_t dataset) {
    if (!c) return BI_ERR_INVALID;
    if (c->pending) return fail(c, BI_ERR_STATE, "bi_eval_begin: the previous bi_eval_begin has not been collected with bi_eval_end");
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (c->d > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    rc = eval_single(c, z, rate_scale, dataset, nullptr, nullptr, /*wait=*/false);
    if (rc) c->pending = 0;
    return rc;
}

int bi_eval_end(bi_ctx* c, double* out, int32_t* status) {
    if (!c || !out) return BI_ERR_INVALID;
    if (!c->pending) return fail(c, BI_ERR_STATE, "bi_eval_end without bi_eval_begin");
    const int kind = c->pending;
    c->pending = 0;
    if (kind == 2) {
        *out = c->pending_ll;
        if (status) *status = c->pending_status;
        return BI_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (status) *status = 0;
    return single_wait(c, c->pending_seq, out, status);
}

int bi_eval_grad(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, double* ll,
                 double* grad, int32_t* status) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (P < 0 || (P > 0 && (!ll || !grad))) return fail(c, BI_ERR_INVALID, "bad P / output pointers");
    if (c->d > 0 && P > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    if (c->bb_source >= 0) {
        HIP_TRY(c, hipSetDevice(c->device));
        return eval_grad_bb(c, P, z, rate_scale, dataset, ll, grad, status);
    }
    const int S = c->S, d = c->d;
    const int W = 1 + d + S;
    if (W > kMaxG) return fail(c, BI_ERR_INVALID, "1 + d + S = %d exceeds %d gradient columns", W, kMaxG);
    HIP_TRY(c, hipSetDevice(c->device));
    const int G = std::max(2, pick_class(W, kMaxG));
    const int de = (int)c->eff_axes.size();
    const int nc = 1 << de, NS = nc * S;
    bool any_neg = false;
    for (int q = 0; q < S; ++q) any_neg |= (c->allow_neg[(size_t)q] != 0);
    const bool unb = c->unbinned;       // extended unbinned likelihood: the rows are pdf values at the events, no counts
    const bool sparse = !unb && c->sparse && c->compact_ready && c->ps_nonneg && !any_neg;
    if (!unb && !sparse && !c->dense_counts) return fail(c, BI_ERR_STATE, "dataset counts are not resident in dense form");
    const int64_t n_rows = c->A * S;
    const double ninf = -std::numeric_limits<double>::infinity();
    const double qnan = std::numeric_limits<double>::quiet_NaN();

    // large batches: the descriptors are built on the device (k_grad_fill), one work item per point
    if (!unb && c->device_plan_min > 0 && P >= c->device_plan_min && de <= 6 && P <= ((int64_t)1 << 26)) {
        // one dataset, up to 32 streams: grouped by grid cell, two matrix products per 16-bin block (k_grad_mfma)
        if (c->grad_mfma && P >= c->grad_mfma_min && c->scan_mfma && c->ps_finite && NS <= 32 && (!dataset || c->T == 1))
            return eval_grad_mfma(c, P, z, rate_scale, dataset, sparse, ll, grad, status);
        return eval_grad_device(c, P, z, rate_scale, dataset, sparse, G, ll, grad, status);
    }

    // Host half, per point and independent: phase 1 decides which points are evaluated at all (the reference's early
    // exits), phase 2 fills the descriptor arrays of the live ones.  Both run on a few host threads for large batches --
    // the batched profile-fit engine calls this once per optimiser iteration over every running problem, and at ~1.7 us
    // per point single-threaded the host half was six times the kernels' time at 10^5 points.
    std::vector<int64_t> corner_off((size_t)nc);
    for (int k = 0; k < nc; ++k) corner_off[(size_t)k] = corner_offset(c, k);
    const std::vector<double> ones((size_t)S, 1.0);
    std::vector<int32_t> st_of((size_t)P, 0);
    parallel_for(P, 2048, [&](int64_t lo, int64_t hi) {
        PointGeom g;
        std::vector<double> r((size_t)S);
        for (int64_t p = lo; p < hi; ++p) {
            ll[p] = ninf;
            for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = qnan;
            const int64_t ds = (dataset && !unb) ? dataset[p] : 0;
            if (!unb && (ds < 0 || ds >= c->T)) { st_of[(size_t)p] = BI_ST_BAD_DATASET; continue; }
            if (!point_geometry(c, z ? z + p * d : nullptr, g)) { st_of[(size_t)p] = BI_ST_OUT_OF_BOUNDS; continue; }
            interp_mus(c, g, r.data());
            const double* rs = rate_scale ? rate_scale + p * S : ones.data();
            for (int s = 0; s < S; ++s) r[(size_t)s] *= rs[s];
            if (!rates_physical(c, r.data())) st_of[(size_t)p] = BI_ST_UNPHYSICAL;
        }
    });
    std::vector<int64_t> live;  // point index of every item
    live.reserve((size_t)P);
    for (int64_t p = 0; p < P; ++p) {
        if (status) status[p] = st_of[(size_t)p];
        if (!st_of[(size_t)p]) live.push_back(p);
    }
    const int64_t n_live = (int64_t)live.size();
    std::vector<int64_t> rowoff((size_t)n_live * NS), cnt_off((size_t)n_live), perm((size_t)n_live * G, -1);
    std::vector<double> coef((size_t)n_live * NS * G, 0.0), slot_lg((size_t)n_live * G, 0.0);
    std::vector<int32_t> tiles((size_t)n_live);
    parallel_for(n_live, 1024, [&](int64_t lo, int64_t hi) {
        PointGeom g;
        std::vector<double> mus((size_t)S), r((size_t)S), dmus((size_t)S * std::max(d, 1)), dw((size_t)nc * std::max(de, 1));
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t p = live[(size_t)i];
            const int64_t ds = (dataset && !unb) ? dataset[p] : 0;
            point_geometry(c, z ? z + p * d : nullptr, g);
            interp_mus(c, g, mus.data());
            const double* rs = rate_scale ? rate_scale + p * S : ones.data();
            for (int s = 0; s < S; ++s) r[(size_t)s] = mus[(size_t)s] * rs[s];
            // d w_c / d z_i for the effective axes: (+-1/delta_i) * prod_{j != i} w^(j)
            for (int corner = 0; corner < nc; ++corner)
                for (int ii = 0; ii < de; ++ii) {
                    const int ax = c->eff_axes[(size_t)ii];
                    double v = (((corner >> (de - 1 - ii)) & 1) ? 1.0 : -1.0) * g.inv_delta[ax];
                    for (int j = 0; j < de; ++j) {
                        if (j == ii) continue;
                        const double t = g.t[c->eff_axes[(size_t)j]];
                        v *= ((corner >> (de - 1 - j)) & 1) ? t : (1 - t);
                    }
                    dw[(size_t)corner * de + ii] = v;
                }
            // d mus_s / d z_i
            for (int ii = 0; ii < de; ++ii)
                for (int s = 0; s < S; ++s) {
                    double v = 0.0;
                    for (int corner = 0; corner < nc; ++corner)
                        v += dw[(size_t)corner * de + ii] * c->h_mus[(size_t)((g.cell_anchor + corner_off[(size_t)corner]) * S + s)];
                    dmus[(size_t)ii * S + s] = v;
                }
            const int64_t row_stride = sparse ? c->h_c_np[(size_t)ds] : c->Bp;
            const int64_t row_base = sparse ? c->h_c_off[(size_t)ds] : 0;
            const size_t ro = (size_t)i * NS, co = (size_t)i * NS * G, po = (size_t)i * G;
            int k = 0;
            for (int corner = 0; corner < nc; ++corner)
                for (int s = 0; s < S; ++s, ++k) {
                    const int64_t row = (g.cell_anchor + corner_off[(size_t)corner]) * S + s;
                    rowoff[ro + k] = row_base + row * row_stride;
                    double* col = &coef[co + (size_t)k * G];
                    const double w = g.w[(size_t)corner];
                    col[0] = w * r[(size_t)s];
                    for (int ii = 0; ii < de; ++ii)   // total derivative w.r.t. z: through the weights and through mus(z)
                        col[1 + c->eff_axes[(size_t)ii]] = dw[(size_t)corner * de + ii] * r[(size_t)s] + w * dmus[(size_t)ii * S + s] * rs[s];
                    col[1 + d + s] = w * mus[(size_t)s];
                    if (sparse) {
                        const double tz = c->h_Tz[(size_t)(ds * n_rows + row)];
                        for (int q = 0; q < W; ++q) slot_lg[po + q] += col[q] * tz;
                    }
                }
            if (unb) {
                // -sum_s mu_s and its derivatives (likelihood.py:690): what the kernel's sums over the events are reduced by
                double rsum = 0.0;
                for (int s = 0; s < S; ++s) rsum += r[(size_t)s];
                slot_lg[po] = rsum;
                for (int ii = 0; ii < de; ++ii) {
                    double v = 0.0;
                    for (int s = 0; s < S; ++s) v += dmus[(size_t)ii * S + s] * rs[s];
                    slot_lg[po + 1 + c->eff_axes[(size_t)ii]] = v;
                }
                for (int s = 0; s < S; ++s) slot_lg[po + 1 + d + s] = mus[(size_t)s];
            } else
                slot_lg[po] += c->h_lgsum[(size_t)ds];
            for (int q = 0; q < W; ++q) perm[po + q] = i * W + q;
            cnt_off[(size_t)i] = unb ? 0 : (sparse ? c->h_cnt_off[(size_t)ds] : ds * c->Bp);
            tiles[(size_t)i] = (int32_t)(row_stride / kTile);
        }
    });
    int max_tiles = 1;
    for (int32_t t : tiles) max_tiles = std::max(max_tiles, (int)t);
    const int64_t n_items = (int64_t)live.size();
    if (n_items == 0) return BI_OK;
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu;
    const int nbx = (int)std::min<int64_t>(max_tiles, n_items == 1 ? slots : std::max<int64_t>(1, (4 * slots + n_items - 1) / n_items));
    // descriptors: one packed copy; results: k_finish writes them straight into pinned host memory
    DevBuf d_part, d_flag;
    auto cleanup = [&]() { dev_free(d_part); dev_free(d_flag); };
    PackedUpload pu;
    const size_t out_bytes = (size_t)n_items * W * sizeof(double);
    if ((rc = packed_upload(c, {{rowoff.data(), rowoff.size() * sizeof(int64_t)}, {coef.data(), coef.size() * sizeof(double)},
                                {cnt_off.data(), cnt_off.size() * sizeof(int64_t)}, {tiles.data(), tiles.size() * sizeof(int32_t)},
                                {perm.data(), perm.size() * sizeof(int64_t)}, {slot_lg.data(), slot_lg.size() * sizeof(double)}},
                            out_bytes, pu)) ||
        (rc = dev_alloc(c, d_part, (size_t)n_items * nbx * G * sizeof(double))) ||
        (rc = dev_alloc(c, d_flag, (size_t)n_items * nbx * G * sizeof(unsigned)))) {
        cleanup();
        return rc;
    }
    double* h_out = (double*)pu.host_out();
    LaunchArgs a{};
    a.ps = sparse ? (const double*)c->ps_c.p : (const double*)c->ps.p;
    a.counts = sparse ? (const double*)c->cnt_c.p : (const double*)c->counts.p;
    a.B = c->B; a.Bp = c->Bp; a.n0 = NS; a.n_tiles = max_tiles; a.chunks = (int)c->tile_chunks;
    a.outlier = c->outlier;
    a.nan_S = (unb && !c->ps_finite) ? c->S : 0;
    if (unb) a.counts = (const double*)c->ps.p;        // (never read in this mode: any valid device address)
    const bool nt = !sparse && (c->nt_loads == 1 || (c->nt_loads == 2 && n_items == 1));
    for (int64_t i0 = 0; i0 < n_items; i0 += 65535) {
        const int64_t ni = std::min<int64_t>(65535, n_items - i0);
        LaunchArgs b = a;
        b.rowoff = pu.dev<int64_t>(0) + i0 * NS;
        b.coef = pu.dev<double>(1) + i0 * NS * G;
        b.item_cnt = pu.dev<int64_t>(2) + i0;
        b.item_tiles = pu.dev<int32_t>(3) + i0;
        b.partial = (double*)d_part.p + i0 * nbx * G;
        b.pflags = (unsigned*)d_flag.p + i0 * nbx * G;
        launch_morph_grad(c, G, b, dim3((unsigned)nbx, (unsigned)ni), nt);
        const int64_t n_slots = ni * G;
        const int lanes = nbx > 64 ? kThreads : 64;
        const int per_block = kThreads / lanes;
        hipLaunchKernelGGL(k_finish, dim3((unsigned)((n_slots + per_block - 1) / per_block)), dim3(kThreads), 0, c->stream,
                           (const double*)b.partial, (const unsigned*)b.pflags, nbx, G, lanes, n_slots,
                           pu.dev<int64_t>(4) + i0 * G, pu.dev<double>(5) + i0 * G, h_out, (int32_t*)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_grad: %s", hipGetErrorString(e));
    for (int64_t i = 0; i < n_items; ++i) {
        const int64_t p = live[(size_t)i];
        ll[p] = h_out[(size_t)i * W];
        for (int j = 0; j < d + S; ++j) grad[p * (d + S) + j] = (unb && !std::isfinite(ll[p])) ? qnan : h_out[(size_t)i * W + 1 + j];
    }
    return BI_OK;
}

// ---- toy-MC form ---------------------------------------------------------------------------

}  // extern "C"

namespace {

// Tile-major copy of the context's non-empty-bin lists for tiles of `tile_bins` bins (a power of two, at most kDotTile): the
// entries of bin tile 0 (dataset 0, 1, ...), then tile 1, ... -- see k_tm_scatter.  Two-byte entries where every count is at
// most 7 (toys of sparse expectations), four-byte entries otherwise; ok = false (and nothing kept) when some count fits neither.
int build_tile_major(bi_ctx* c, int tile_bins, DevBuf& entries, DevBuf& off, bool& ok, int& width_out) {
    int tile_shift = 0;
    while ((1 << tile_shift) < tile_bins) ++tile_shift;
    const int n_tl = (int)((c->B + tile_bins - 1) / tile_bins);
    const int64_t nnz = c->h_nz_off.back(), cells = (int64_t)n_tl * c->T;
    const uint32_t pad_entry = (uint32_t)tile_bins << 3;          // four-byte lists: the offset of the extra LDS slot behind the tile (0.0), count 0
    int rc;
    DevBuf d_tile, d_cnt, d_tmp, d_bad;
    auto drop = [&]() { dev_free(d_tile); dev_free(d_cnt); dev_free(d_tmp); dev_free(d_bad); };
    size_t scan_bytes = 0;
    (void)prim_exclusive_scan_sum(nullptr, scan_bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)(cells + 1), c->stream);
    if ((rc = dev_alloc(c, d_tile, (size_t)c->T * (n_tl + 1) * sizeof(int32_t))) || (rc = dev_alloc(c, d_cnt, (size_t)(cells + 1) * sizeof(int64_t))) ||
        (rc = dev_alloc(c, d_tmp, std::max<size_t>(scan_bytes, 256))) || (rc = dev_alloc(c, d_bad, 64)) ||
        (rc = dev_alloc(c, off, (size_t)(cells + 1) * sizeof(int64_t)))) {
        drop();
        return rc;
    }
    hipLaunchKernelGGL(k_csr_tile_offsets, dim3((unsigned)c->T), dim3(kThreads), 0, c->stream, (const int32_t*)c->nz_idx.p,
                       (const int64_t*)c->nz_off.p, n_tl, (int32_t*)d_tile.p, tile_shift);
    hipError_t e = hipGetLastError();
    // two-byte entries where every count is at most 7 (toys of sparse expectations), four-byte entries otherwise: the first
    // format is tried, and a count that does not fit sends the build to the second
    ok = false;
    for (int width = c->dot_entry16 ? 2 : 4; e == hipSuccess && !ok; width = 4) {
        const int group = 16 / width;
        const size_t n_entries = (size_t)(nnz + (group - 1) * cells + kDotPad);     // (every run padded to whole 16-byte groups; + the kernel's read-ahead)
        if ((rc = dev_alloc(c, entries, n_entries * width))) { drop(); return rc; }
        e = hipMemsetAsync(d_bad.p, 0, 64, c->stream);
        // (two-byte entries: 13 bits of offset -- tiles of 8192 bins use them all, their padding is entry 0, dropped by a select on
        //  the count; tiles of up to 4096 bins leave the bit for the offset of the extra zero slot behind the tile, as four-byte
        //  entries have it)
        if (e == hipSuccess && width == 2 && tile_bins > 4096) e = hipMemsetAsync(entries.p, 0, n_entries * 2, c->stream);
        if (e == hipSuccess && width == 2 && tile_bins <= 4096) e = hipMemsetD16Async((hipDeviceptr_t)entries.p, (unsigned short)pad_entry, n_entries, c->stream);
        if (e == hipSuccess && width == 4) e = hipMemsetD32Async((hipDeviceptr_t)entries.p, (int)pad_entry, n_entries, c->stream);
        hipLaunchKernelGGL(k_tm_counts, dim3((unsigned)((cells + 1 + 255) / 256)), dim3(256), 0, c->stream, (const int32_t*)d_tile.p, c->T,
                           n_tl, (int64_t*)d_cnt.p, group);
        size_t tb = d_tmp.bytes;
        if (e == hipSuccess) e = prim_exclusive_scan_sum(d_tmp.p, tb, (const int64_t*)d_cnt.p, (int64_t*)off.p, (int64_t)0, (size_t)(cells + 1), c->stream);
        if (width == 2)
            hipLaunchKernelGGL((k_tm_scatter<uint16_t>), dim3((unsigned)c->T), dim3(kThreads), 0, c->stream, (const int32_t*)c->nz_idx.p,
                               (const double*)c->nz_n.p, (const int64_t*)c->nz_off.p, (const int32_t*)d_tile.p, c->T, n_tl,
                               (const int64_t*)off.p, (uint16_t*)entries.p, (int*)d_bad.p, tile_shift);
        else
            hipLaunchKernelGGL((k_tm_scatter<uint32_t>), dim3((unsigned)c->T), dim3(kThreads), 0, c->stream, (const int32_t*)c->nz_idx.p,
                               (const double*)c->nz_n.p, (const int64_t*)c->nz_off.p, (const int32_t*)d_tile.p, c->T, n_tl,
                               (const int64_t*)off.p, (uint32_t*)entries.p, (int*)d_bad.p, tile_shift);
        int bad = 0;
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) break;
        ok = bad == 0;
        width_out = width;
        if (width == 4) break;
    }
    if (e != hipSuccess) (void)hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_datasets (tile-major lists): %s", hipGetErrorString(e));
    if (!ok) { dev_free(entries); dev_free(off); }
    return BI_OK;
}

// out_dev != NULL: results stay in HBM at out_dev[0 .. t1 - t0) (the call still returns after the stream has drained:
// its descriptors travel through the context's pinned staging block, which the next call reuses)
int eval_datasets_impl(bi_ctx* c, const double* z, const double* rate_scale, int64_t t0, int64_t t1, double* out,
                       double* out_dev, int32_t* status) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (c->bb_source >= 0) return fail(c, BI_ERR_INVALID, "bi_eval_datasets is not available with Beeston-Barlow");
    if (c->unbinned) return fail(c, BI_ERR_INVALID, "bi_eval_datasets needs a binned likelihood");
    if (t0 < 0 || t1 > c->T || t0 > t1) return fail(c, BI_ERR_INVALID, "dataset range [%lld,%lld) outside [0,%lld)", (long long)t0, (long long)t1, (long long)c->T);
    if (c->d > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    if (t1 > t0 && !out && !out_dev) return fail(c, BI_ERR_INVALID, "out is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t n = t1 - t0;
    if (status) *status = 0;
    const double ninf = -std::numeric_limits<double>::infinity();
    auto reject = [&](int32_t bit) -> int {          // the reference returns -inf before it looks at any data
        if (status) *status = bit;
        if (out_dev) {
            // on the context's stream, like every other write to (and the caller's collectives on) that buffer
            if (n) {
                hipLaunchKernelGGL(k_fill_value, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, out_dev, n, ninf);
                hipError_t e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_datasets_device: %s", hipGetErrorString(e));
            }
        } else {
            std::fill(out, out + n, ninf);
        }
        return BI_OK;
    };
    PointGeom g;
    if (!point_geometry(c, z, g)) return reject(BI_ST_OUT_OF_BOUNDS);
    std::vector<double> r((size_t)c->S);
    interp_mus(c, g, r.data());
    if (rate_scale) for (int s = 0; s < c->S; ++s) r[(size_t)s] *= rate_scale[s];
    if (!rates_physical(c, r.data())) return reject(BI_ST_UNPHYSICAL);
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
    DevBuf d_out;
    auto cleanup = [&]() { dev_free(d_out); };
    const bool csr = (c->sparse && c->csr_ready) || !c->dense_counts;
    if (csr && !c->csr_ready) return fail(c, BI_ERR_STATE, "no counts resident");
    // Non-empty-bin lists with enough entries per (dataset, bin tile): the tiled kernel (log mu staged through LDS, the
    // lists re-ordered tile by tile once per data upload).
    const int n_tl = (int)((c->B + kDotTile - 1) / kDotTile);
    bool tiled = csr && c->dot_tiled && n >= 64 && n_tl >= 4 && c->h_nz_off.size() == (size_t)c->T + 1 &&
                 c->h_nz_off.back() >= (int64_t)16 * c->T * n_tl && (int64_t)c->T * (n_tl + 1) <= ((int64_t)1 << 28);
    if (tiled && c->nz_tile_epoch != c->epoch) {
        if ((rc = build_tile_major(c, kDotTile, c->tm_entries, c->tm_off, c->tm_ok, c->tm_width))) return rc;
        c->nz_tile_epoch = c->epoch;
    }
    tiled = tiled && c->tm_ok;
    const int64_t chunk = tiled ? std::max<int64_t>(64, ((int64_t)1 << 27) / n_tl) : (csr ? 1048576 : 16384);
    const int nbx = tiled ? n_tl : csr ? 1 : (int)std::min<int64_t>(n_tiles, std::max<int64_t>(1, 4 * slots / std::max<int64_t>(1, (std::min(n, chunk) + kDotGroup - 1) / kDotGroup)));
    // descriptors in one packed copy; up to 4 MB of results are written straight into pinned host memory
    const bool host_out = !out_dev && (size_t)n * sizeof(double) <= ((size_t)4 << 20);
    // (up to kMaxSingleStreams streams the point's descriptors ride in the kernel arguments: no copy ahead of the launch)
    const bool by_value = NS <= kMaxSingleStreams && (c->toy_fast_call & 1);
    PackedUpload pu;
    std::vector<std::pair<const void*, size_t>> parts;
    if (!by_value) parts = {{rowoff.data(), rowoff.size() * sizeof(int64_t)}, {coef.data(), coef.size() * sizeof(double)}};
    if ((rc = packed_upload(c, parts, host_out ? (size_t)n * sizeof(double) : 0, pu)) ||
        (rc = dev_alloc(c, c->logmu, (size_t)c->Bp * sizeof(double))) ||
        (rc = dev_alloc(c, c->scratch, (size_t)nmu * sizeof(double) + (size_t)nmu * sizeof(unsigned) + 64)) ||
        (rc = dev_alloc(c, c->scratch2, (size_t)std::min(n, chunk) * nbx * sizeof(double))) ||
        (!host_out && !out_dev && (rc = dev_alloc(c, d_out, (size_t)std::max<int64_t>(n, 1) * sizeof(double))))) {
        cleanup();
        return rc;
    }
    double* res = out_dev ? out_dev : (host_out ? (double*)pu.host_out() : (double*)d_out.p);
    LaunchArgs a{};
    a.ps = (const double*)c->ps.p;
    a.rowoff = by_value ? nullptr : pu.dev<int64_t>(0);
    a.coef = by_value ? nullptr : pu.dev<double>(1);
    a.partial = (double*)c->scratch.p;
    a.pflags = (unsigned*)((char*)c->scratch.p + (((size_t)nmu * sizeof(double) + 63) / 64) * 64);
    a.B = c->B; a.Bp = c->Bp; a.n0 = NS; a.n_tiles = n_tiles;
    // One chunk of tiled partials whose results go to pinned host memory: the finish kernel's last block publishes a
    // sequence number behind them, and the call polls that word (as single evaluations do) instead of synchronising
    unsigned long long* done_word = nullptr;
    unsigned long long seq = 0;
    if (tiled && host_out && n <= chunk && c->poll_result && !c->profiling && (c->toy_fast_call & 6) == 6) {
        if ((rc = dev_alloc(c, c->toy_blocks_done, 64))) { cleanup(); return rc; }
        if (!c->toy_blocks_done_zeroed) {
            HIP_TRY(c, hipMemsetAsync(c->toy_blocks_done.p, 0, 64, c->stream));
            c->toy_blocks_done_zeroed = true;
        }
        done_word = (unsigned long long*)((char*)pu.host_out() + ((size_t)n * sizeof(double) + 63) / 64 * 64);   // (packed_upload keeps 64 spare bytes behind the results)
        seq = ++c->toy_seq;
        *(volatile unsigned long long*)done_word = 0ull;
    }
    auto launch_logmu = [&]() {
        EventScope ev(c);
        if (by_value) {
            PointDesc pd;
            memcpy(pd.rowoff, rowoff.data(), rowoff.size() * sizeof(int64_t));
            memcpy(pd.coef, coef.data(), coef.size() * sizeof(double));
            hipLaunchKernelGGL(k_morph_logmu_desc, dim3((unsigned)nmu), dim3(kThreads), 0, c->stream, a, pd, (double*)c->logmu.p, 0);
        } else
            hipLaunchKernelGGL(k_morph_logmu, dim3((unsigned)nmu), dim3(kThreads), 0, c->stream, a, (double*)c->logmu.p, 0);
    };
    launch_logmu();
    for (int64_t s0 = 0; s0 < n; s0 += chunk) {
        const int64_t ni = std::min(chunk, n - s0);
        {
            EventScope ev(c);
            if (tiled) {
                // datasets split over blockIdx.y so that ~4 blocks per CU exist; every block stages its tile once
                // (and few enough datasets per block for its 32-bit entry indices: datasets x kDotTile < 2^31)
                // as many blocks as are resident at once (the 8-lane variant's registers allow one per CU, the 16-lane
                // variant's two; asked from the runtime once): ONE round, every block's start-up paid once
                // variant: 4-byte entries <8, 3> (96 slots per run) or <16, 2>; 2-byte entries <8, 2, 2> (128 slots) or, dot_lanes 4,
                // <4, 3, 2> (96 slots)
                // (dot_lanes 0 = the best measured for the entry width: 8 lanes with four-byte entries, 4 with two-byte ones)
                const int variant = c->tm_width == 2 ? ((c->dot_lanes == 4 || c->dot_lanes == 0) ? 3 : 2) : (c->dot_lanes == 16 ? 1 : 0);
                static std::atomic<int> resident_cache[4];
                std::atomic<int>& rslot = resident_cache[variant];
                int resident = c->dot_blocks_per_cu > 0 ? (int)c->dot_blocks_per_cu : rslot.load();
                if (resident <= 0) {
                    int r = 0;
                    const void* f = variant == 0 ? (const void*)k_dataset_dot_tiled<8, 3, 4> : variant == 1 ? (const void*)k_dataset_dot_tiled<16, 2, 4>
                                  : variant == 2 ? (const void*)k_dataset_dot_tiled<8, 2, 2> : (const void*)k_dataset_dot_tiled<4, 3, 2>;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, f, kDotThreads, 0) != hipSuccess || r < 1) r = 1;
                    rslot.store(r);
                    resident = r;
                }
                const unsigned by = (unsigned)std::max<int64_t>({1, std::min<int64_t>((ni + 255) / 256, (int64_t)resident * c->prop.multiProcessorCount / n_tl),
                                                                 (ni + 262143) / 262144});
#define BI_DOT(L, AHEAD, W)                                                                                        \
    hipLaunchKernelGGL((k_dataset_dot_tiled<L, AHEAD, W>), dim3((unsigned)n_tl, by), dim3(kDotThreads), 0, c->stream, \
                       (const void*)c->tm_entries.p, (const int64_t*)c->tm_off.p, c->T, n_tl,                      \
                       (const double*)c->logmu.p, c->B, t0 + s0, ni, (double*)c->scratch2.p)
                if (variant == 0) BI_DOT(8, 3, 4); else if (variant == 1) BI_DOT(16, 2, 4); else if (variant == 2) BI_DOT(8, 2, 2); else BI_DOT(4, 3, 2);
#undef BI_DOT
            } else if (csr)
                hipLaunchKernelGGL(k_dataset_dot_csr, dim3((unsigned)ni), dim3(kThreads), 0, c->stream,
                                   (const int32_t*)c->nz_idx.p, (const double*)c->nz_n.p, (const int64_t*)c->nz_off.p,
                                   (const double*)c->logmu.p, t0 + s0, (double*)c->scratch2.p);
            else
                hipLaunchKernelGGL(k_dataset_dot, dim3((unsigned)nbx, (unsigned)((ni + kDotGroup - 1) / kDotGroup)), dim3(kThreads), 0,
                                   c->stream, (const double*)c->counts.p, (const double*)c->logmu.p, c->Bp, n_tiles, t0 + s0, ni,
                                   (double*)c->scratch2.p);
        }
        if (tiled && (c->toy_fast_call & 2))
            hipLaunchKernelGGL(k_dataset_finish_tiled, dim3((unsigned)((ni + 63) / 64)), dim3(kThreads), 0, c->stream,
                               (const double*)c->scratch2.p, nbx, (const double*)a.partial, (const unsigned*)a.pflags, nmu,
                               (const double*)c->lgsum.p, t0 + s0, ni, res + s0, (unsigned*)c->toy_blocks_done.p, done_word, seq);
        else
            hipLaunchKernelGGL(k_dataset_finish, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, c->stream,
                               (const double*)c->scratch2.p, nbx, tiled ? (int64_t)1 : (int64_t)nbx, tiled ? ni : (int64_t)1,
                               (const double*)a.partial, (const unsigned*)a.pflags, nmu,
                               (const double*)c->lgsum.p, t0 + s0, ni, res + s0);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && n && !host_out && !out_dev) e = hipMemcpyAsync(out, d_out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    bool arrived = false;
    if (e == hipSuccess && done_word) {
        // The word arrives ~0.1 ms after the launches for 10^4 datasets: spin for the first ~30 us (a yield costs more than
        // the wait is worth there), then give the core away between looks -- eight ranks of a node each burning a core per
        // call is what the launcher's thread cap exists to prevent -- and fall back to the stream synchronise (which also
        // reports a faulted kernel) after a time that scales with the work: 2 ms + 1 us per dataset, at most 50 ms.
        const volatile unsigned long long* dw = done_word;
        const auto t_start = std::chrono::steady_clock::now();
        const auto t_spin = t_start + std::chrono::microseconds(30);
        const auto t_end = t_start + std::chrono::microseconds(std::min<int64_t>(50000, 2000 + n));
        bool yielding = false;
        for (unsigned spin = 0; !(arrived = (*dw == seq)); ++spin) {
            if (yielding) {
                sched_yield();
                if (std::chrono::steady_clock::now() > t_end) break;
                continue;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spin & 63u) == 63u && std::chrono::steady_clock::now() > t_spin) yielding = true;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        ++c->n_toy_polled;
    }
    // (falls back to the stream sync, which also reports a faulted kernel; every 256th call synchronises anyway, so the
    //  runtime retires its completed commands at a steady pace)
    if (e == hipSuccess && (!arrived || (seq & 255ull) == 0)) e = hipStreamSynchronize(c->stream);
    // (a completion word that never came -- a launch that failed part way -- leaves the finish kernel's block counter in an
    //  unknown state: it is zeroed again before the next call, which would otherwise wait out its 50 ms every time)
    if (done_word && !arrived) c->toy_blocks_done_zeroed = false;
    if (e == hipSuccess && n && host_out) memcpy(out, res, (size_t)n * sizeof(double));
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_eval_datasets: %s", hipGetErrorString(e));
    return BI_OK;
}

}  // namespace

#include "bi_toy_points.h"

extern "C" {

int bi_eval_datasets_points(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, int64_t t0, int64_t t1, double* out,
                            int32_t* status) {
    if (!c) return BI_ERR_INVALID;
    return eval_datasets_points_impl(c, P, z, rate_scale, t0, t1, out, nullptr, status);
}

int bi_eval_datasets_points_device(bi_ctx* c, int64_t P, const double* z, const double* rate_scale, int64_t t0, int64_t t1,
                                   double* out_dev, int32_t* status) {
    if (!c) return BI_ERR_INVALID;
    if (P > 0 && t1 > t0 && !out_dev) return fail(c, BI_ERR_INVALID, "out_dev is NULL");
    return eval_datasets_points_impl(c, P, z, rate_scale, t0, t1, nullptr, out_dev, status);
}

int bi_eval_datasets(bi_ctx* c, const double* z, const double* rate_scale, int64_t t0, int64_t t1, double* out,
                     int32_t* status) {
    return eval_datasets_impl(c, z, rate_scale, t0, t1, out, nullptr, status);
}

int bi_eval_datasets_device(bi_ctx* c, const double* z, const double* rate_scale, int64_t t0, int64_t t1, double* out_dev,
                            int32_t* status) {
    if (t1 > t0 && !out_dev) return fail(c, BI_ERR_INVALID, "out_dev is NULL");
    return eval_datasets_impl(c, z, rate_scale, t0, t1, nullptr, out_dev, status);
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
    DevBuf d_row, d_coef, d_cnt, d_off, d_lgp, d_p0;
    auto cleanup = [&]() { dev_free(d_row); dev_free(d_coef); dev_free(d_cnt); dev_free(d_off); dev_free(d_lgp); dev_free(d_p0); };
    if ((rc = dev_upload(c, d_row, rowoff)) || (rc = dev_upload(c, d_coef, coef)) ||
        (rc = dev_alloc(c, c->logmu, (size_t)c->Bp * sizeof(double))) || (rc = dev_alloc(c, d_p0, (size_t)c->Bp * sizeof(double))) ||
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
    const double* p0 = (const double*)d_p0.p;
    hipLaunchKernelGGL(k_exp_neg, dim3((unsigned)((c->Bp + 255) / 256)), dim3(256), 0, c->stream, mu, c->Bp, (double*)d_p0.p);
    // Sparse expectations (M = sum mu << B, templates and rates >= 0): event by event -- N ~ Poisson(M), the bins by bisection
    // in the cumulative sums, sorted and run-length encoded per toy (k_toy_events); else one draw per bin.
    c->last_toy_method = 0;
    if (c->toy_events && c->ps_nonneg && B >= 4096) {
        DevBuf d_cdf, d_tmp, d_nev, d_room, d_nnz, d_ovf, d_tidx, d_tn;
        auto drop = [&]() { dev_free(d_cdf); dev_free(d_tmp); dev_free(d_nev); dev_free(d_room); dev_free(d_nnz); dev_free(d_ovf); dev_free(d_tidx); dev_free(d_tn); };
        size_t sb1 = 0, sb2 = 0;
        (void)prim_inclusive_scan_sum(nullptr, sb1, (const double*)nullptr, (double*)nullptr, (size_t)B, c->stream);
        (void)prim_exclusive_scan_sum(nullptr, sb2, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)(T + 1), c->stream);
        if ((rc = dev_alloc(c, d_cdf, (size_t)B * sizeof(double))) || (rc = dev_alloc(c, d_tmp, std::max<size_t>({sb1, sb2, 256}))) ||
            (rc = dev_alloc(c, d_ovf, 64))) { drop(); cleanup(); return rc; }
        size_t tb = d_tmp.bytes;
        hipError_t e = prim_inclusive_scan_sum(d_tmp.p, tb, mu, (double*)d_cdf.p, (size_t)B, c->stream);
        double M = 0.0;
        if (e == hipSuccess) e = hipMemcpyAsync(&M, (const double*)d_cdf.p + (B - 1), sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_ovf.p, 0, 64, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { drop(); cleanup(); return fail(c, BI_ERR_HIP, "toy generation (cumulative sums): %s", hipGetErrorString(e)); }
        const double bound = M + 12.0 * std::sqrt(std::max(M, 1.0)) + 32.0;
        int npow2 = 1024;
        while (npow2 < bound && npow2 < 65536) npow2 <<= 1;
        if (M > 0.0 && M == M && M < (double)B / 8.0 && npow2 <= 32768) {
            const size_t lds = (npow2 <= 16384 ? (size_t)2 * npow2 * sizeof(uint32_t) + 16 * kEvThreads * sizeof(uint16_t)
                                               : (size_t)npow2 * sizeof(uint32_t)) + kEvThreads * (sizeof(int) + sizeof(double)) + 64;
            if ((e = hipFuncSetAttribute((const void*)k_toy_events, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) {
                drop(); cleanup();
                return fail(c, BI_ERR_HIP, "toy generation (LDS size of the event kernel): %s", hipGetErrorString(e));
            }
            if ((rc = dev_alloc(c, d_nev, (size_t)(T + 1) * sizeof(int64_t))) || (rc = dev_alloc(c, d_room, (size_t)(T + 1) * sizeof(int64_t))) ||
                (rc = dev_alloc(c, d_nnz, (size_t)(T + 1) * sizeof(int64_t))) || (rc = dev_alloc(c, c->nz_off, (size_t)(T + 1) * sizeof(int64_t)))) {
                drop(); cleanup(); return rc;
            }
            // (1) events per toy -> room in the provisional lists
            hipLaunchKernelGGL(k_toy_event_counts, dim3((unsigned)((T + 1 + 255) / 256)), dim3(256), 0, c->stream, M, seed, c->toy_offset, T,
                               npow2, (int64_t*)d_nev.p, (int*)d_ovf.p);
            tb = d_tmp.bytes;
            e = prim_exclusive_scan_sum(d_tmp.p, tb, (const int64_t*)d_nev.p, (int64_t*)d_room.p, (int64_t)0, (size_t)(T + 1), c->stream);
            int64_t n_events = 0;
            int ovf = 0;
            if (e == hipSuccess) e = hipMemcpyAsync(&n_events, (const int64_t*)d_room.p + T, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&ovf, d_ovf.p, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { drop(); cleanup(); return fail(c, BI_ERR_HIP, "toy generation (events per toy): %s", hipGetErrorString(e)); }
            if (!ovf) {
                if ((rc = dev_alloc(c, d_tidx, (size_t)std::max<int64_t>(n_events, 1) * sizeof(int32_t))) ||
                    (rc = dev_alloc(c, d_tn, (size_t)std::max<int64_t>(n_events, 1) * sizeof(double)))) { drop(); cleanup(); return rc; }
                // (2) one block per toy: events -> sorted bins -> (bin, count) runs, written into the toy's room
                const int64_t tchunk_ev = 65535;
                for (int64_t t0 = 0; t0 < T; t0 += tchunk_ev) {
                    const int64_t n = std::min(tchunk_ev, T - t0);
                    hipLaunchKernelGGL(k_toy_events, dim3((unsigned)n), dim3(kEvThreads), lds, c->stream, (const double*)d_cdf.p, B, M, seed,
                                       t0 + c->toy_offset, npow2, (const int64_t*)d_room.p + t0, (int32_t*)d_tidx.p, (double*)d_tn.p,
                                       (int64_t*)d_nnz.p + t0, (double*)c->lgsum.p + t0);
                }
                // (3) non-empty bins per toy -> final offsets; pack
                e = hipMemsetAsync((int64_t*)d_nnz.p + T, 0, sizeof(int64_t), c->stream);
                tb = d_tmp.bytes;
                if (e == hipSuccess) e = prim_exclusive_scan_sum(d_tmp.p, tb, (const int64_t*)d_nnz.p, (int64_t*)c->nz_off.p, (int64_t)0, (size_t)(T + 1), c->stream);
                c->h_nz_off.assign((size_t)T + 1, 0);
                if (e == hipSuccess) e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(c->h_nz_off.data(), c->nz_off.p, (size_t)(T + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) { drop(); cleanup(); return fail(c, BI_ERR_HIP, "toy generation (events): %s", hipGetErrorString(e)); }
                const int64_t run = c->h_nz_off[(size_t)T];
                if ((rc = dev_alloc(c, c->nz_idx, (size_t)std::max<int64_t>(run, 1) * sizeof(int32_t))) ||
                    (rc = dev_alloc(c, c->nz_n, (size_t)std::max<int64_t>(run, 1) * sizeof(double)))) { drop(); cleanup(); return rc; }
                for (int64_t t0 = 0; t0 < T; t0 += tchunk_ev) {
                    const int64_t n = std::min(tchunk_ev, T - t0);
                    hipLaunchKernelGGL(k_toy_pack, dim3((unsigned)n), dim3(kThreads), 0, c->stream, (const int64_t*)d_room.p + t0,
                                       (const int64_t*)c->nz_off.p + t0, (const int32_t*)d_tidx.p, (const double*)d_tn.p,
                                       (int32_t*)c->nz_idx.p, (double*)c->nz_n.p);
                }
                c->h_lgsum.assign((size_t)T, 0.0);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(c->h_lgsum.data(), c->lgsum.p, (size_t)T * sizeof(double), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                drop();
                cleanup();
                if (e != hipSuccess) return fail(c, BI_ERR_HIP, "toy generation (events, pack): %s", hipGetErrorString(e));
                c->last_toy_method = 1;
                c->T = T;
                c->csr_ready = true;
                if ((rc = build_compact_templates(c))) return rc;
                c->data_ready = true;
                return BI_OK;
            }
        }
        drop();            // not sparse enough, or a toy beyond the sort buffer (12 sigma): one draw per bin below
    }
    const int64_t tchunk = 32768;
    for (int64_t t0 = 0; t0 < T; t0 += tchunk) {
        const int64_t n = std::min(tchunk, T - t0);
        hipLaunchKernelGGL(k_toy_count, dim3((unsigned)nchunks, (unsigned)n), dim3(kThreads), 0, c->stream, mu, p0, B, seed,
                           t0 + c->toy_offset, (int32_t*)d_cnt.p + t0 * nchunks, nchunks);
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
        hipLaunchKernelGGL(k_toy_scatter, dim3((unsigned)nchunks, (unsigned)n), dim3(kThreads), 0, c->stream, mu, p0, B, seed,
                           t0 + c->toy_offset, (const int64_t*)d_off.p + t0 * nchunks, nchunks, (int32_t*)c->nz_idx.p, (double*)c->nz_n.p,
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


int bi_counts_to_dense(bi_ctx* c) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (c->dense_counts) return BI_OK;
    if (!c->csr_ready) return fail(c, BI_ERR_STATE, "no counts resident");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->T * c->Bp * sizeof(double);
    if ((rc = dev_alloc(c, c->counts, bytes))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->counts.p, 0, bytes, c->stream));
    for (int64_t t = 0; t < c->T; ++t) {
        const int64_t lo = c->h_nz_off[(size_t)t], nnz = c->h_nz_off[(size_t)t + 1] - lo;
        if (nnz > 0)
            hipLaunchKernelGGL(k_csr_to_dense, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, c->stream,
                               (const int32_t*)c->nz_idx.p + lo, (const double*)c->nz_n.p + lo, nnz, (double*)c->counts.p + t * c->Bp);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->dense_counts = true;
    ++c->epoch;                                  // plans made over the lists alone are stale
    return BI_OK;
}


// ---- extended unbinned likelihood -----------------------------------------------------------------

int bi_set_unbinned(bi_ctx* c, double outlier_likelihood) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (c->bb_source >= 0) return fail(c, BI_ERR_INVALID, "Beeston-Barlow applies to binned likelihoods only");
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

}  // extern "C"

namespace {

// coords: host [k][N], or coords_dev: the same block already in HBM (bi_simulate_events)
int score_events_impl(bi_ctx* tp, bi_ctx* c, int method, int k, const int32_t* n_grid, const double* grid, int64_t N,
                      const double* coords, const double* coords_dev, double outlier_likelihood) {
    if (!c) return BI_ERR_INVALID;
    if (!tp || tp == c) return fail(c, BI_ERR_INVALID, "need a templates context different from the target");
    if (c->pending || tp->pending) return fail(c, BI_ERR_STATE, "a bi_eval_begin is outstanding: call bi_eval_end first");
    if (!tp->model_ready) return fail(c, BI_ERR_STATE, "the templates context holds no model");
    if (tp->device != c->device) return fail(c, BI_ERR_INVALID, "templates and target live on different devices");
    if (tp->bb_source >= 0) return fail(c, BI_ERR_INVALID, "Beeston-Barlow applies to binned likelihoods only");
    if (method != 0 && method != 1) return fail(c, BI_ERR_INVALID, "method must be 0 (piecewise) or 1 (linear)");
    if (k < 1 || k > kMaxDim || !n_grid || !grid) return fail(c, BI_ERR_INVALID, "need 1..%d axes with grid values", kMaxDim);
    if (N < 0 || (N > 0 && !coords && !coords_dev)) return fail(c, BI_ERR_INVALID, "bad N / coords");
    ScoreArgs a{};
    a.k = k;
    a.method = method;
    a.clip = coords_dev ? 1 : 0;          // events simulated on the device arrive unclipped; a caller's events are clipped already
    int64_t bins = 1;
    int off = 0;
    for (int i = 0; i < k; ++i) {
        if (n_grid[i] < 2) return fail(c, BI_ERR_INVALID, "axis %d needs at least two grid values", i);
        for (int j = 1; j < n_grid[i]; ++j)
            if (!(grid[off + j] > grid[off + j - 1])) return fail(c, BI_ERR_INVALID, "grid values of axis %d are not strictly ascending", i);
        a.n_grid[i] = n_grid[i];
        a.grid_off[i] = off;
        off += n_grid[i];
        bins *= method == 0 ? n_grid[i] - 1 : n_grid[i];
    }
    if (bins != tp->B) return fail(c, BI_ERR_INVALID, "the grid describes %lld bins, the templates have %lld", (long long)bins, (long long)tp->B);
    int64_t step = 1;
    for (int i = k - 1; i >= 0; --i) { a.stride[i] = step; step *= method == 0 ? n_grid[i] - 1 : n_grid[i]; }
    // the target becomes a model on the same anchor grid with one "bin" per event
    std::vector<int32_t> na(tp->n_anchor.begin(), tp->n_anchor.end());
    std::vector<double> az;
    for (int i = 0; i < tp->d; ++i) az.insert(az.end(), tp->grid[(size_t)i].begin(), tp->grid[(size_t)i].end());
    int rc = bi_model_begin(c, tp->d, na.data(), az.data(), tp->S, N, -1);
    if (rc) return rc;
    if (N > 0) {
        DevBuf d_ev, d_grid, d_base, d_t, d_keys, d_iota, d_tmp;
        auto drop = [&]() { dev_free(d_ev); dev_free(d_grid); dev_free(d_base); dev_free(d_t); dev_free(d_keys); dev_free(d_iota); dev_free(d_tmp); };
        // events ordered by cell (see k_score_rows): from a few thousand events on, and while 32-bit positions do
        const bool sorted = c->score_sorted && N >= 4096 && N < ((int64_t)1 << 31);
        size_t sort_bytes = 0;
        if (sorted) (void)prim_sort_pairs(nullptr, sort_bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr,
                                                    (int32_t*)nullptr, (size_t)N, 0u, 64u, c->stream);
        if ((!coords_dev && (rc = dev_alloc(c, d_ev, (size_t)N * k * sizeof(double)))) || (rc = dev_alloc(c, d_grid, (size_t)off * sizeof(double))) ||
            (rc = dev_alloc(c, d_base, (size_t)N * sizeof(int64_t))) || (method == 1 && (rc = dev_alloc(c, d_t, (size_t)N * k * sizeof(double)))) ||
            (sorted && ((rc = dev_alloc(c, d_keys, (size_t)N * sizeof(int64_t))) || (rc = dev_alloc(c, d_iota, (size_t)N * sizeof(int32_t))) ||
                        (rc = dev_alloc(c, d_tmp, std::max<size_t>(sort_bytes, 256))) || (rc = dev_alloc(c, c->ev_perm, (size_t)N * sizeof(int32_t)))))) {
            drop();
            return rc;
        }
        hipError_t e = coords_dev ? hipSuccess : hipMemcpyAsync(d_ev.p, coords, (size_t)N * k * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_grid.p, grid, (size_t)off * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(tp->stream);            // whatever filled the templates is complete
        if (e == hipSuccess) {
            // every event's cell and weights once, then the gathers row by row: the row is the slow dimension of the grid, so
            // the chip works on one or two 8 MB histograms at a time (see k_score_rows)
            const int n_rows = (int)(tp->A * tp->S);
            hipLaunchKernelGGL(k_score_locate, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                               coords_dev ? coords_dev : (const double*)d_ev.p, N, a, (const double*)d_grid.p, (int64_t*)d_base.p, (double*)d_t.p);
            const int64_t* base_used = (const int64_t*)d_base.p;
            if (sorted) {
                hipLaunchKernelGGL(k_iota32, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (int32_t*)d_iota.p, N);
                size_t tb = d_tmp.bytes;
                // (the cell index needs ceil(log2 B) bits: fewer radix passes than 64)
                unsigned bits = 1;
                while (bits < 63 && ((int64_t)1 << bits) < tp->B) ++bits;
                e = prim_sort_pairs(d_tmp.p, tb, (const int64_t*)d_base.p, (int64_t*)d_keys.p, (const int32_t*)d_iota.p,
                                              (int32_t*)c->ev_perm.p, (size_t)N, 0u, bits, c->stream);
                base_used = (const int64_t*)d_keys.p;
            }
            const int per_block = kThreads * score_events_per_thread(method == 0 ? 0 : k);
            const unsigned bx = (unsigned)((N + per_block - 1) / per_block);
#define BI_ROWS(K)                                                                                                 \
    hipLaunchKernelGGL((k_score_rows<K>), dim3(bx, (unsigned)std::min(n_rows, 65535)), dim3(kThreads), 0, c->stream, \
                       base_used, (const double*)d_t.p, N, a, (const double*)tp->ps.p, tp->Bp, n_rows, (double*)c->ps.p, c->Bp,        \
                       sorted ? (const int32_t*)c->ev_perm.p : (const int32_t*)nullptr)
            if (e == hipSuccess) switch (method == 0 ? 0 : k) {
                case 0: BI_ROWS(0); break; case 1: BI_ROWS(1); break; case 2: BI_ROWS(2); break; case 3: BI_ROWS(3); break;
                case 4: BI_ROWS(4); break; case 5: BI_ROWS(5); break; case 6: BI_ROWS(6); break; case 7: BI_ROWS(7); break;
                default: BI_ROWS(8); break;
            }
#undef BI_ROWS
            if (e == hipSuccess) e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);              // coords are borrowed for the call only
        else (void)hipStreamSynchronize(c->stream);
        drop();
        c->ev_sorted = e == hipSuccess && sorted;
        if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_score_events: %s", hipGetErrorString(e));
    }
    c->h_mus = tp->h_mus;
    std::fill(c->anchor_set.begin(), c->anchor_set.end(), 1);
    if ((rc = bi_model_end(c))) return rc;
    c->allow_neg = tp->allow_neg;
    return bi_set_unbinned(c, outlier_likelihood);
}

}  // namespace

extern "C" {

int bi_score_events(bi_ctx* tp, bi_ctx* c, int method, int k, const int32_t* n_grid, const double* grid, int64_t N,
                    const double* coords, double outlier_likelihood) {
    return score_events_impl(tp, c, method, k, n_grid, grid, N, coords, nullptr, outlier_likelihood);
}

int bi_simulate_events(bi_ctx* tp, bi_ctx* c, const double* z, const double* rate_scale, int method, int k, const int32_t* n_edges,
                       const double* edges, uint64_t seed, double outlier_likelihood, int64_t* n_per_source) {
    if (!c) return BI_ERR_INVALID;
    if (!tp || tp == c) return fail(c, BI_ERR_INVALID, "need a templates context different from the target");
    if (c->pending || tp->pending) return fail(c, BI_ERR_STATE, "a bi_eval_begin is outstanding: call bi_eval_end first");
    if (!tp->model_ready) return fail(c, BI_ERR_STATE, "the templates context holds no model");
    if (tp->device != c->device) return fail(c, BI_ERR_INVALID, "templates and target live on different devices");
    if (method != 0 && method != 1) return fail(c, BI_ERR_INVALID, "method must be 0 (piecewise) or 1 (linear)");
    if (k < 1 || k > kMaxDim || !n_edges || !edges) return fail(c, BI_ERR_INVALID, "need 1..%d axes with bin edges", kMaxDim);
    if (tp->d > 0 && !z) return fail(c, BI_ERR_INVALID, "z is NULL");
    SimArgs a{};
    a.k = k; a.S = tp->S;
    int64_t bins = 1;
    int off = 0;
    for (int i = 0; i < k; ++i) {
        if (n_edges[i] < (method == 1 ? 3 : 2)) return fail(c, BI_ERR_INVALID, "axis %d has too few bin edges", i);
        for (int j = 1; j < n_edges[i]; ++j)
            if (!(edges[off + j] > edges[off + j - 1])) return fail(c, BI_ERR_INVALID, "bin edges of axis %d are not strictly ascending", i);
        a.n_edges[i] = n_edges[i];
        a.edge_off[i] = off;
        off += n_edges[i];
        bins *= n_edges[i] - 1;
    }
    if (bins != tp->B) return fail(c, BI_ERR_INVALID, "the edges describe %lld bins, the templates have %lld", (long long)bins, (long long)tp->B);
    int64_t step = 1;
    for (int i = k - 1; i >= 0; --i) { a.stride[i] = step; step *= n_edges[i] - 1; }
    HIP_TRY(c, hipSetDevice(c->device));
    // expected events per source at z (the scalar half of likelihood.py:355-393), then N_s ~ Poisson
    PointGeom g;
    if (!point_geometry(tp, z, g)) return fail(c, BI_ERR_INVALID, "simulation point is outside the anchor box");
    const int S = tp->S;
    std::vector<double> r((size_t)S);
    interp_mus(tp, g, r.data());
    if (rate_scale) for (int s = 0; s < S; ++s) r[(size_t)s] *= rate_scale[s];
    for (int s = 0; s < S; ++s)
        if (!(r[(size_t)s] >= 0.0 && r[(size_t)s] < std::numeric_limits<double>::infinity()))
            return fail(c, BI_ERR_INVALID, "event simulation needs rates in [0, inf)");
    const int nc = (int)g.w.size();
    const int64_t B = tp->B;
    std::vector<int64_t> rowoff((size_t)S * nc);
    for (int s = 0; s < S; ++s)
        for (int corner = 0; corner < nc; ++corner)
            rowoff[(size_t)s * nc + corner] = ((g.cell_anchor + corner_offset(tp, corner)) * S + s) * tp->Bp;
    DevBuf d_row, d_w, d_dens, d_cdf, d_edges, d_rates, d_n, d_first, d_tmp;
    auto cleanup = [&]() { dev_free(d_row); dev_free(d_w); dev_free(d_dens); dev_free(d_cdf); dev_free(d_edges); dev_free(d_rates);
                           dev_free(d_n); dev_free(d_first); dev_free(d_tmp); };
    int rc;
    size_t scan_bytes = 0;
    (void)prim_inclusive_scan_sum(nullptr, scan_bytes, (const double*)nullptr, (double*)nullptr, (size_t)B, c->stream);
    std::vector<double> h_edges(edges, edges + off);
    if ((rc = dev_upload(c, d_row, rowoff)) || (rc = dev_upload(c, d_w, g.w)) || (rc = dev_upload(c, d_edges, h_edges)) ||
        (rc = dev_upload(c, d_rates, r)) || (rc = dev_alloc(c, d_dens, (size_t)S * B * sizeof(double))) ||
        (rc = dev_alloc(c, d_cdf, (size_t)S * B * sizeof(double))) || (rc = dev_alloc(c, d_n, (size_t)S * sizeof(int64_t))) ||
        (rc = dev_alloc(c, d_tmp, std::max<size_t>(scan_bytes, 256)))) { cleanup(); return rc; }
    hipError_t e = hipStreamSynchronize(tp->stream);                      // whatever filled the templates is complete
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_morph_store, dim3((unsigned)((B + kThreads - 1) / kThreads), (unsigned)S), dim3(kThreads), 0, c->stream,
                           (const double*)tp->ps.p, (const int64_t*)d_row.p, (const double*)d_w.p, nc, B, (double*)d_dens.p);
        hipLaunchKernelGGL(k_sim_pmf, dim3((unsigned)((B + kThreads - 1) / kThreads), (unsigned)S), dim3(kThreads), 0, c->stream,
                           (const double*)d_dens.p, a, (const double*)d_edges.p, B, (double*)d_dens.p);
        e = hipGetLastError();
    }
    for (int s = 0; e == hipSuccess && s < S; ++s) {
        size_t tb = d_tmp.bytes;
        e = prim_inclusive_scan_sum(d_tmp.p, tb, (const double*)d_dens.p + (size_t)s * B, (double*)d_cdf.p + (size_t)s * B, (size_t)B, c->stream);
    }
    std::vector<int64_t> n_s((size_t)S, 0);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sim_counts, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, c->stream, (const double*)d_rates.p, S, seed, (int64_t*)d_n.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(n_s.data(), d_n.p, (size_t)S * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return fail(c, BI_ERR_HIP, "bi_simulate_events: %s", hipGetErrorString(e)); }
    std::vector<int64_t> first((size_t)S + 1, 0);
    for (int s = 0; s < S; ++s) first[(size_t)s + 1] = first[(size_t)s] + n_s[(size_t)s];
    const int64_t N = first[(size_t)S];
    if (n_per_source) std::copy(n_s.begin(), n_s.end(), n_per_source);
    // the events themselves: coordinates [k][N] and the source of every event, kept with the target for bi_download_events
    if ((rc = dev_alloc(c, c->sim_coords, (size_t)std::max<int64_t>(N, 1) * k * sizeof(double))) ||
        (rc = dev_alloc(c, c->sim_source, (size_t)std::max<int64_t>(N, 1) * sizeof(int32_t))) || (rc = dev_upload(c, d_first, first))) { cleanup(); return rc; }
    c->sim_k = k;
    c->sim_n = N;
    if (N > 0) {
        hipLaunchKernelGGL(k_sim_events, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, (const double*)d_cdf.p, B, a,
                           (const double*)d_edges.p, (const int64_t*)d_first.p, seed, N, (double*)c->sim_coords.p, (int32_t*)c->sim_source.p);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    cleanup();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_simulate_events: %s", hipGetErrorString(e));
    // score them at every anchor model: the grid of the lookup is the edges ('piecewise') or the bin centres ('linear')
    std::vector<int32_t> n_grid((size_t)k);
    std::vector<double> grid;
    int eo = 0;
    for (int i = 0; i < k; ++i) {
        if (method == 0) {
            n_grid[(size_t)i] = n_edges[i];
            grid.insert(grid.end(), edges + eo, edges + eo + n_edges[i]);
        } else {
            n_grid[(size_t)i] = n_edges[i] - 1;
            for (int j = 0; j + 1 < n_edges[i]; ++j) grid.push_back(0.5 * (edges[eo + j] + edges[eo + j + 1]));
        }
        eo += n_edges[i];
    }
    // (bi_model_begin inside re-allocates the target's model, not the sim_* buffers)
    return score_events_impl(tp, c, method, k, n_grid.data(), grid.data(), N, nullptr, (const double*)c->sim_coords.p, outlier_likelihood);
}

int bi_download_events(bi_ctx* c, double* coords, int32_t* source) {
    if (!c) return BI_ERR_INVALID;
    if (c->sim_n < 0) return fail(c, BI_ERR_STATE, "no simulated events are resident (bi_simulate_events first)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->sim_n > 0 && coords)
        HIP_TRY(c, hipMemcpyAsync(coords, c->sim_coords.p, (size_t)c->sim_n * c->sim_k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (c->sim_n > 0 && source)
        HIP_TRY(c, hipMemcpyAsync(source, c->sim_source.p, (size_t)c->sim_n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BI_OK;
}

int64_t bi_simulated_event_count(const bi_ctx* c) { return c ? c->sim_n : -1; }

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
                       (const int64_t*)d_row.p, (const double*)d_w.p, nc, c->B, (double*)d_out.p,
                       // (an unbinned tensor scored on the device holds its events ordered by cell: back to the caller's order)
                       which == 0 && c->ev_sorted ? (const int32_t*)c->ev_perm.p : (const int32_t*)nullptr);
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
    if (c->bb_exact == 1 || (c->bb_exact == 2 && bb_zero_u_possible(c, g, mus_out))) {
        if ((rc = bb_exact_total(c, g, &Ntot))) return rc;     // as the evaluation above did
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

// ---- plain device buffers (gather staging for multi-GPU runs) -----------------------------------

int bi_device_alloc(bi_ctx* c, int64_t bytes, void** out) {
    if (!c || !out || bytes < 0) return BI_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) {                       // the context's parked buffers (up to 4 GiB) go first
        drop_recycle_cache(c);
        e = hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16));
    }
    if (e != hipSuccess) return fail(c, BI_ERR_NOMEM, "hipMalloc(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(e));
    c->user_allocs.push_back(*out);           // whatever is still alive goes with the context (bi_destroy)
    return BI_OK;
}

int bi_device_free(bi_ctx* c, void* p) {
    if (!c) return BI_ERR_INVALID;
    if (!p) return BI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    auto it = std::find(c->user_allocs.begin(), c->user_allocs.end(), p);
    if (it == c->user_allocs.end()) return fail(c, BI_ERR_INVALID, "bi_device_free: %p was not allocated by bi_device_alloc on this context", p);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->user_allocs.erase(it);
    HIP_TRY(c, hipFree(p));
    return BI_OK;
}

// Copies of up to 4 MB between a device buffer and pageable host memory go through a pinned bounce buffer of the context:
// the runtime's own handling of pageable memory costs 15 ... 110 us per small copy (it varies from call to call), the
// bounce a DMA plus a memcpy -- the per-call gather of 10^4 toy results (80 KB) is such a copy.
constexpr int64_t kBounceBytes = (int64_t)4 << 20;

static int bounce_of(bi_ctx* c) {
    if (!c->bounce_host) HIP_TRY(c, hipHostMalloc(&c->bounce_host, (size_t)kBounceBytes, hipHostMallocDefault));
    return BI_OK;
}

int bi_memcpy_to_host(bi_ctx* c, void* dst, const void* src, int64_t bytes) {
    if (!c || bytes < 0 || (bytes > 0 && (!dst || !src))) return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (bytes > 0 && bytes <= kBounceBytes) {
        int rc = bounce_of(c);
        if (rc) return rc;
        if (bytes % 4 == 0 && ((uintptr_t)src & 3) == 0) {      // a kernel writing through the host mapping (see k_copy_words)
            const int64_t n_words = bytes / 4;
            hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<int64_t>(1024, (n_words + 255) / 256)), dim3(256), 0, c->stream,
                               (const uint32_t*)src, (uint32_t*)c->bounce_host, n_words);
            HIP_TRY(c, hipGetLastError());
        } else {
            HIP_TRY(c, hipMemcpyAsync(c->bounce_host, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        memcpy(dst, c->bounce_host, (size_t)bytes);
        return BI_OK;
    }
    if (bytes) HIP_TRY(c, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BI_OK;
}

int bi_memcpy_to_device(bi_ctx* c, void* dst, const void* src, int64_t bytes) {
    if (!c || bytes < 0 || (bytes > 0 && (!dst || !src))) return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (bytes > 0 && bytes <= kBounceBytes) {
        int rc = bounce_of(c);
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));          // an earlier copy out of the bounce buffer is complete
        memcpy(c->bounce_host, src, (size_t)bytes);
        HIP_TRY(c, hipMemcpyAsync(dst, c->bounce_host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return BI_OK;
    }
    if (bytes) HIP_TRY(c, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BI_OK;
}

// ---- self-tests of device math ---------------------------------------------------------------

int bi_selftest_log(bi_ctx* c, int64_t n, const double* x, double* out) {
    if (!c || n < 0 || (n > 0 && (!x || !out))) return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf dx, dy;
    int rc;
    if ((rc = dev_alloc(c, dx, (size_t)std::max<int64_t>(n, 1) * sizeof(double))) ||
        (rc = dev_alloc(c, dy, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) { dev_free(dx); dev_free(dy); return rc; }
    hipError_t e = n ? hipMemcpyAsync(dx.p, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream) : hipSuccess;
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_selftest_log, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double*)dx.p, n, (double*)dy.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess && n) e = hipMemcpyAsync(out, dy.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(dx); dev_free(dy);
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_selftest_log: %s", hipGetErrorString(e));
    return BI_OK;
}

// the library's own device-wide primitives (tu_prim.hip) on caller data: kind of sort 0 (uint64 keys, int64 values), 1 (int64, int32),
// 2 (double, int32), 3 (the counting sort: uint64 keys below `end_bit` <= 1024, int64 values); kind of scan 0 inclusive max int64, 1 inclusive sum int64, 2 inclusive sum double, 3 exclusive sum int64 (+ init)
int bi_selftest_sort(bi_ctx* c, int kind, int64_t n, const void* keys, const void* vals, int begin_bit, int end_bit, void* keys_out, void* vals_out) {
    if (!c || n < 0 || kind < 0 || kind > 3 || begin_bit < 0 || end_bit > (kind == 3 ? 1024 : 64) || begin_bit > end_bit || (kind == 3 && end_bit < 1) ||
        (n > 0 && (!keys || !vals || !keys_out || !vals_out)))
        return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t vb = (kind == 0 || kind == 3) ? 8 : 4, nn = (size_t)std::max<int64_t>(n, 1);
    DevBuf dk, dv, dk2, dv2, dt;
    auto drop = [&]() { dev_free(dk); dev_free(dv); dev_free(dk2); dev_free(dv2); dev_free(dt); };
    size_t tb = 0;
    if (kind == 0) (void)prim_sort_pairs(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int64_t*)nullptr, (int64_t*)nullptr, (size_t)n, 0u, 64u, c->stream);
    else if (kind == 3) (void)prim_count_sort_pairs(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int64_t*)nullptr, (int64_t*)nullptr, (size_t)n, (uint64_t)end_bit, c->stream);
    else if (kind == 1) (void)prim_sort_pairs(nullptr, tb, (const int64_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0u, 64u, c->stream);
    else (void)prim_sort_pairs(nullptr, tb, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0u, 64u, c->stream);
    int rc;
    if ((rc = dev_alloc(c, dk, nn * 8)) || (rc = dev_alloc(c, dv, nn * vb)) || (rc = dev_alloc(c, dk2, nn * 8)) || (rc = dev_alloc(c, dv2, nn * vb)) ||
        (rc = dev_alloc(c, dt, std::max<size_t>(tb, 256)))) { drop(); return rc; }
    hipError_t e = n ? hipMemcpyAsync(dk.p, keys, (size_t)n * 8, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    if (e == hipSuccess && n) e = hipMemcpyAsync(dv.p, vals, (size_t)n * vb, hipMemcpyHostToDevice, c->stream);
    size_t t2 = dt.bytes;
    if (e == hipSuccess) {
        if (kind == 0) e = prim_sort_pairs(dt.p, t2, (const uint64_t*)dk.p, (uint64_t*)dk2.p, (const int64_t*)dv.p, (int64_t*)dv2.p, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, c->stream);
        else if (kind == 3) e = prim_count_sort_pairs(dt.p, t2, (const uint64_t*)dk.p, (uint64_t*)dk2.p, (const int64_t*)dv.p, (int64_t*)dv2.p, (size_t)n, (uint64_t)end_bit, c->stream);
        else if (kind == 1) e = prim_sort_pairs(dt.p, t2, (const int64_t*)dk.p, (int64_t*)dk2.p, (const int32_t*)dv.p, (int32_t*)dv2.p, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, c->stream);
        else e = prim_sort_pairs(dt.p, t2, (const double*)dk.p, (double*)dk2.p, (const int32_t*)dv.p, (int32_t*)dv2.p, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, c->stream);
    }
    if (e == hipSuccess && n) e = hipMemcpyAsync(keys_out, dk2.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(vals_out, dv2.p, (size_t)n * vb, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream); else (void)hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_selftest_sort: %s", hipGetErrorString(e));
    return BI_OK;
}

int bi_selftest_scan(bi_ctx* c, int kind, int64_t n, const void* in, int64_t init, void* out) {
    if (!c || n < 0 || kind < 0 || kind > 3 || (n > 0 && (!in || !out))) return BI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    DevBuf di, dout, dt;
    auto drop = [&]() { dev_free(di); dev_free(dout); dev_free(dt); };
    size_t tb = 0;
    (void)prim_exclusive_scan_sum(nullptr, tb, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)n, c->stream);   // (all kinds: 8-byte elements, the same size)
    int rc;
    if ((rc = dev_alloc(c, di, nn * 8)) || (rc = dev_alloc(c, dout, nn * 8)) || (rc = dev_alloc(c, dt, std::max<size_t>(tb, 256)))) { drop(); return rc; }
    hipError_t e = n ? hipMemcpyAsync(di.p, in, (size_t)n * 8, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    size_t t2 = dt.bytes;
    if (e == hipSuccess) {
        if (kind == 0) e = prim_inclusive_scan_max(dt.p, t2, (const int64_t*)di.p, (int64_t*)dout.p, (size_t)n, c->stream);
        else if (kind == 1) e = prim_inclusive_scan_sum(dt.p, t2, (const int64_t*)di.p, (int64_t*)dout.p, (size_t)n, c->stream);
        else if (kind == 2) e = prim_inclusive_scan_sum(dt.p, t2, (const double*)di.p, (double*)dout.p, (size_t)n, c->stream);
        else e = prim_exclusive_scan_sum(dt.p, t2, (const int64_t*)di.p, (int64_t*)dout.p, init, (size_t)n, c->stream);
    }
    if (e == hipSuccess && n) e = hipMemcpyAsync(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream); else (void)hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_selftest_scan: %s", hipGetErrorString(e));
    return BI_OK;
}

// ---- measurement ---------------------------------------------------------------------------

int bi_measure_read_bandwidth(bi_ctx* c, int nontemporal, int blocks_per_cu, int reps, double* gb_per_s) {
    if (!c || !gb_per_s || reps < 1 || blocks_per_cu < 1) return BI_ERR_INVALID;
    if (!c->model_ready || !c->ps.p) return fail(c, BI_ERR_STATE, "bi_measure_read_bandwidth: no model resident");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf sink;
    int rc = dev_alloc(c, sink, 8);
    if (rc) return rc;
    const int64_t n2 = c->A * c->S * c->Bp / 2;       // the whole template tensor, in 16-byte elements
    const dim3 grid((unsigned)(c->prop.multiProcessorCount * blocks_per_cu));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0.0;
    for (int r = 0; e == hipSuccess && r <= reps; ++r) {      // the first pass is the warm-up
        e = hipEventRecord(e0, c->stream);
        if (nontemporal) hipLaunchKernelGGL(k_read_sum<true>, grid, dim3(kThreads), 0, c->stream, (const double*)c->ps.p, n2, (double*)sink.p);
        else hipLaunchKernelGGL(k_read_sum<false>, grid, dim3(kThreads), 0, c->stream, (const double*)c->ps.p, n2, (double*)sink.p);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && r > 0 && ms > 0.f) best = std::max(best, (double)n2 * 16.0 / (ms * 1e6));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    dev_free(sink);
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_measure_read_bandwidth: %s", hipGetErrorString(e));
    *gb_per_s = best;
    return BI_OK;
}

int bi_measure_stream_bandwidth(bi_ctx* c, int items, int rows, int nontemporal, int blocks_per_cu, int reps, double* gb_per_s) {
    if (!c || !gb_per_s || reps < 1 || blocks_per_cu < 1 || items < 1 || rows < 1) return BI_ERR_INVALID;
    if (!c->model_ready || !c->ps.p) return fail(c, BI_ERR_STATE, "bi_measure_stream_bandwidth: no model resident");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf sink;
    int rc = dev_alloc(c, sink, 8);
    if (rc) return rc;
    const int64_t total_rows = c->A * c->S;
    // same launch shape as a batched morph launch: blocks_per_cu resident blocks per CU over all items; both the morph
    // kernel's own 512-bin tiles and (where the padded row length allows) 1024-bin tiles
    const int64_t slots = (int64_t)c->prop.multiProcessorCount * blocks_per_cu;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0.0;
    for (int pieces = 1; pieces <= 2; ++pieces) {
        if (c->Bp % (kTile * pieces)) continue;
        const int n_tiles = (int)(c->Bp / (kTile * pieces));
        const int nbx = (int)std::min<int64_t>(n_tiles, std::max<int64_t>(1, slots / items));
        const dim3 grid((unsigned)nbx, (unsigned)items);
        for (int r = 0; e == hipSuccess && r <= reps; ++r) {      // the first pass is the warm-up; every pass starts elsewhere
            const int64_t first = ((int64_t)r * items * rows * 7) % total_rows;
            e = hipEventRecord(e0, c->stream);
#define BI_ROWS(NTv, Pv) hipLaunchKernelGGL((k_read_rows<NTv, Pv>), grid, dim3(kThreads), 0, c->stream, (const double*)c->ps.p, c->Bp, total_rows, first, rows, n_tiles, (int)c->tile_chunks, (double*)sink.p)
            if (nontemporal) { if (pieces == 1) BI_ROWS(true, 1); else BI_ROWS(true, 2); }
            else { if (pieces == 1) BI_ROWS(false, 1); else BI_ROWS(false, 2); }
#undef BI_ROWS
            if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e == hipSuccess && r > 0 && ms > 0.f)
                best = std::max(best, (double)items * rows * (double)c->Bp * 8.0 / (ms * 1e6));
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    dev_free(sink);
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_measure_stream_bandwidth: %s", hipGetErrorString(e));
    *gb_per_s = best;
    return BI_OK;
}

int bi_measure_copy_bandwidth(bi_ctx* c, int64_t bytes, int reps, double* gb_per_s) {
    if (!c || !gb_per_s || reps < 1 || bytes < 1) return BI_ERR_INVALID;
    if (!c->model_ready || !c->ps.p) return fail(c, BI_ERR_STATE, "bi_measure_copy_bandwidth: no model resident");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)std::min<int64_t>(bytes, c->A * c->S * c->Bp * (int64_t)sizeof(double));
    void* dst = nullptr;
    hipError_t e = hipMalloc(&dst, n);
    if (e != hipSuccess) return fail(c, BI_ERR_NOMEM, "bi_measure_copy_bandwidth: %s", hipGetErrorString(e));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0.0;
    for (int r = 0; e == hipSuccess && r <= reps; ++r) {      // the first pass is the warm-up
        e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dst, c->ps.p, n, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && r > 0 && ms > 0.f) best = std::max(best, 2.0 * (double)n / (ms * 1e6));   // bytes read + bytes written
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dst);
    if (e != hipSuccess) return fail(c, BI_ERR_HIP, "bi_measure_copy_bandwidth: %s", hipGetErrorString(e));
    *gb_per_s = best;
    return BI_OK;
}

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

#include "bi_fit.h"
