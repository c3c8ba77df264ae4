// host_backend.cpp -- libblueice_host.so: the minimal entry points of include/blueice_hip.h as plain C++ loops on the
// host, in the reference's own operation order.
//
// WHAT THIS IS FOR.  SURVEY.md section 7 step 3: "a pure-C++ CPU backend behind the same ABI (lets every API test run
// without a GPU)".  It exists so that the BOUNDARY can be exercised where there is no GPU: the reference-side binding of
// INTEGRATION.md (tools/reference_stub/hip_backend.py) runs the reference's own test files over these entry points in the
// development container (tools/run_reference_tests_over_stub.py), and the C caller of examples/c_abi_demo.c runs in the CPU
// test suite.  It is NOT a fallback of the product: nothing in blueice_amd/ loads it (blueice_amd/_capi.py binds
// libblueice_hip.so only, and DeviceContext raises DeviceError without a GPU), it is not linked into the GPU library, and
// bench.py never touches it.  It is not the oracle either (oracle/ is numpy/scipy); the two are independent restatements
// that tests/test_host_backend.py holds against each other and against the reference's golden fixtures.
//
// Entry points (the section 8(b) minimal set): bi_create / bi_destroy / bi_last_error / bi_version / bi_device_info,
// bi_upload_model, bi_model_begin / bi_model_set_anchor / bi_model_end, bi_set_allow_negative, bi_upload_counts, bi_eval,
// bi_eval_full, bi_interpolate; bi_set_param / bi_get_param / bi_list_params exist and know no parameter.
//
// Arithmetic, with the reference lines each loop follows (JelleAalbers/blueice v1.2.1; scipy 1.15.3 / numpy 2.2 where the
// reference calls into them):
//   morph        scipy RegularGridInterpolator._evaluate_linear behind blueice/pdf_morphers.py:67-70: corners in
//                itertools.product order (axis 0 slowest), weight ((1 w_0) w_1)..., value = value + V w from 0.0
//   rates        blueice/likelihood.py:355,366-393 (one combined scale per source, as bi_eval's rate_scale argument)
//   early exits  :345-347 (outside the anchor box -> -inf), :397-415 (unphysical rates -> -inf)
//   Beeston-Barlow  :618-660 and the root formulas :693-712, term by term in Python's evaluation order
//   Poisson      :662-675: rows scaled, summed over sources in sequence, scipy.stats.poisson.logpmf's argument handling,
//                (xlogy(n, mu) - gammaln(n + 1)) - mu with cephes' lgam (published algorithm, restated below), np.sum
//   np.sum       numpy's pairwise summation: chunks of 8192 elements in sequence, each chunk halved down to blocks of <= 128
//                elements summed in 8 strided accumulators
// Compiled with -ffp-contract=off: every multiplication and addition rounds on its own, as numpy's do.
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/blueice_hip.h"

namespace {

const double kInf = std::numeric_limits<double>::infinity();
const double kNaN = std::numeric_limits<double>::quiet_NaN();
std::string g_create_error;

}  // namespace

struct bi_ctx {
    std::string err;
    int d = 0;
    std::vector<int> n_anchor;
    std::vector<std::vector<double>> grid;
    int64_t A = 0;
    int S = 0;
    int64_t B = 0;
    int bb_source = -1;
    bool model_ready = false, streaming = false;
    std::vector<char> anchor_set;
    std::vector<double> ps;      // [A][S][B]
    std::vector<double> mus;     // [A][S]
    std::vector<double> nm;      // [A][B]   row bb_source of n_model_events
    std::vector<int> allow_negative;
    int64_t T = 0;
    std::vector<double> counts;  // [T][B]
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

// ---- numpy's summation order ------------------------------------------------------------------------------------------
double pairwise(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise(a, n2) + pairwise(a + n2, n - n2);
}

double numpy_sum(const double* a, int64_t n) {
    if (n == 0) return 0.0;
    double total = pairwise(a, std::min<int64_t>(8192, n));
    for (int64_t s = 8192; s < n; s += 8192) total = total + pairwise(a + s, std::min<int64_t>(8192, n - s));
    return total;
}

// ---- cephes lgam for x >= 1 (scipy.special.gammaln's algorithm; the arguments here are n + 1 with n >= 0) ----------------
double polevl(double x, const double* c, int n) {
    double a = c[0];
    for (int i = 1; i <= n; ++i) a = a * x + c[i];
    return a;
}
double p1evl(double x, const double* c, int n) {
    double a = x + c[0];
    for (int i = 1; i < n; ++i) a = a * x + c[i];
    return a;
}
double cephes_lgam(double x) {
    static const double A[] = {8.11614167470508450300E-4, -5.95061904284301438324E-4, 7.93650340457716943945E-4,
                               -2.77777777730099687205E-3, 8.33333333333331927722E-2};
    static const double Bc[] = {-1.37825152569120859100E3, -3.88016315134637840924E4, -3.31612992738871184744E5,
                                -1.16237097492762307383E6, -1.72173700820839662146E6, -8.53555664245765465627E5};
    static const double C[] = {-3.51815701436523470549E2, -1.70642106651881159223E4, -2.20528590553854454839E5,
                               -1.13933444367982507207E6, -2.53252307177582951285E6, -2.01889141433532773231E6};
    const double LS2PI = 0.91893853320467274178, MAXLGM = 2.556348e305;
    if (!(x == x)) return x;
    if (std::isinf(x)) return kInf;
    if (x < 13.0) {
        double z = 1.0, p = 0.0, u = x;
        while (u >= 3.0) { p -= 1.0; u = x + p; z *= u; }
        while (u < 2.0) {
            if (u == 0.0) return kInf;
            z /= u; p += 1.0; u = x + p;
        }
        if (z < 0.0) z = -z;
        if (u == 2.0) return std::log(z);
        p -= 2.0;
        x = x + p;
        p = x * polevl(x, Bc, 5) / p1evl(x, C, 6);
        return std::log(z) + p;
    }
    if (x > MAXLGM) return kInf;
    double q = (x - 0.5) * std::log(x) - x + LS2PI;
    if (x > 1.0e8) return q;
    const double p = 1.0 / (x * x);
    if (x >= 1000.0)
        q += ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p + 0.0833333333333333333333) / x;
    else
        q += polevl(p, A, 4) / x;
    return q;
}

// scipy.stats.poisson(mu).logpmf(k): _distn_infrastructure.py logpmf + _discrete_distns.py poisson._logpmf
double poisson_logpmf(double k, double mu) {
    const bool cond0 = mu >= 0.0;                                   // _argcheck
    if (!cond0 || k != k) return kNaN;
    if (!(k >= 0.0) || std::floor(k) != k) return -kInf;           // support / integrality
    const double xl = (k == 0.0) ? 0.0 : k * std::log(mu);          // xlogy (mu is not nan here)
    return (xl - cephes_lgam(k + 1.0)) - mu;
}

// ---- scipy's grid cell and corner weights ----------------------------------------------------------------------------------
struct Cell {
    std::vector<int64_t> off;   // [2^d] linear anchor index of every corner, itertools.product order
    std::vector<double> w;      // [2^d] weights ((1 w_0) w_1) ...
};

bool in_box(const bi_ctx* c, const double* z) {
    for (int i = 0; i < c->d; ++i) {
        const std::vector<double>& g = c->grid[i];
        if (!(g.front() <= z[i] && z[i] <= g.back())) return false;
    }
    return true;
}

Cell find_cell(const bi_ctx* c, const double* z) {
    const int d = c->d;
    std::vector<int> k(d);
    std::vector<double> t(d);
    for (int i = 0; i < d; ++i) {
        const std::vector<double>& g = c->grid[i];
        const int n = (int)g.size();
        if (n == 1) { k[i] = 0; t[i] = 0.0; continue; }
        int kk;
        if (z[i] == g[n - 1]) kk = n - 2;
        else {
            kk = 0;
            while (kk + 1 < n && g[kk + 1] <= z[i]) ++kk;            // searchsorted(side='right') - 1
            if (kk > n - 2) kk = n - 2;
        }
        k[i] = kk;
        t[i] = (z[i] - g[kk]) / (g[kk + 1] - g[kk]);
    }
    std::vector<int64_t> stride(d);
    int64_t s = 1;
    for (int i = d - 1; i >= 0; --i) { stride[i] = s; s *= c->n_anchor[i]; }
    Cell cell;
    const int nc = 1 << d;
    cell.off.resize(nc);
    cell.w.resize(nc);
    for (int corner = 0; corner < nc; ++corner) {
        int64_t off = 0;
        double w = 1.0;
        for (int i = 0; i < d; ++i) {
            const int up = (corner >> (d - 1 - i)) & 1;              // axis 0 slowest
            const int idx = k[i] + ((up && c->n_anchor[i] > 1) ? 1 : 0);
            off += idx * stride[i];
            w = w * (up ? t[i] : 1.0 - t[i]);
        }
        cell.off[corner] = off;
        cell.w[corner] = w;
    }
    return cell;
}

// value = value + V[corner] * weight, from zeros (pdf_morphers.py:67-70 -> scipy _evaluate_linear)
void morph(const Cell& cell, const double* tensor, int64_t row_len, double* out) {
    for (int64_t j = 0; j < row_len; ++j) out[j] = 0.0;
    for (size_t q = 0; q < cell.off.size(); ++q) {
        const double* v = tensor + cell.off[q] * row_len;
        const double w = cell.w[q];
        for (int64_t j = 0; j < row_len; ++j) out[j] = out[j] + v[j] * w;
    }
}

// likelihood.py:397-415
bool rates_physical(const bi_ctx* c, const std::vector<double>& mus) {
    bool any_allowed = false;
    for (int s = 0; s < c->S; ++s) any_allowed |= c->allow_negative[s] != 0;
    if (!any_allowed) {
        for (double m : mus)
            if (!(m >= 0.0 && m < kInf)) return false;
        return true;
    }
    bool any_finite = false;
    for (double m : mus) any_finite |= m < kInf;
    if (!any_finite || numpy_sum(mus.data(), (int64_t)mus.size()) < 0.0) return false;
    for (int s = 0; s < c->S; ++s)
        if (!(0.0 <= mus[s]) && !c->allow_negative[s]) return false;
    return true;
}

// beeston_barlow_root1 / root2 (likelihood.py:693-712), every operation in Python's order
void bb_roots(double a, double p, double U, double d, double* r1, double* r2) {
    const double U2 = U * U, p2 = p * p, a2 = a * a, d2 = d * d;
    double disc = U2 * p2;
    disc = disc + (2.0 * U2) * p;
    disc = disc + U2;
    disc = disc + ((2.0 * U) * a) * p2;
    disc = disc + ((2.0 * U) * a) * p;
    disc = disc - ((2.0 * U) * d) * p2;
    disc = disc - ((2.0 * U) * d) * p;
    disc = disc + a2 * p2;
    disc = disc + ((2.0 * a) * d) * p2;
    disc = disc + d2 * p2;
    const double root = std::sqrt(disc);
    const double head = (((-U) * p - U) + a * p) + d * p;
    const double denom = (2.0 * p) * (p + 1.0);
    *r1 = (head - root) / denom;
    *r2 = (head + root) / denom;
}

struct Eval {
    double ll = 0.0;
    int32_t status = 0;
    std::vector<double> mus;    // [S] adjusted
    std::vector<double> ps;     // [S][B] adjusted
};

// one evaluation of LogLikelihoodBase.__call__ after the settings have been resolved (likelihood.py:345-422)
void evaluate(const bi_ctx* c, const double* z, const double* scale, int64_t dataset, bool keep, Eval* e) {
    const int S = c->S;
    const int64_t B = c->B;
    e->status = 0;
    if (dataset < 0 || dataset >= c->T) { e->status = BI_ST_BAD_DATASET; e->ll = -kInf; return; }
    if (!in_box(c, z)) { e->status = BI_ST_OUT_OF_BOUNDS; e->ll = -kInf; return; }
    const Cell cell = find_cell(c, z);
    std::vector<double> mus(S);
    morph(cell, c->mus.data(), S, mus.data());
    if (scale)
        for (int s = 0; s < S; ++s) mus[s] *= scale[s];
    if (!rates_physical(c, mus)) { e->status = BI_ST_UNPHYSICAL; e->ll = -kInf; return; }
    std::vector<double> ps((size_t)S * B);
    morph(cell, c->ps.data(), (int64_t)S * B, ps.data());
    const double* n = c->counts.data() + dataset * B;

    if (c->bb_source >= 0) {                                        // adjust_expectations, 'bb_single' (:618-660)
        const int si = c->bb_source;
        std::vector<double> a(B), u(B), w(B), A(B), aw(B);
        morph(cell, c->nm.data(), B, a.data());
        for (int64_t b = 0; b < B; ++b) {                           // u_bins = np.sum(counts_per_bin, axis=0), rows in sequence
            double acc = 0.0;
            for (int s = 0; s < S; ++s) {
                const double x = ps[(size_t)s * B + b] * (s != si ? mus[s] : 0.0);
                acc = (s == 0) ? x : acc + x;
            }
            u[b] = acc;
        }
        const double N = numpy_sum(a.data(), B);
        const double p_cal = mus[si] / N;
        for (int64_t b = 0; b < B; ++b) w[b] = ps[(size_t)si * B + b] / a[b] * N;
        for (int64_t b = 0; b < B; ++b) {
            double r1, r2;
            bb_roots(a[b], w[b] * p_cal, u[b], n[b], &r1, &r2);
            if (!(r1 <= 0.0)) e->status |= BI_ST_BB_ROOT1;          // assert np.all(A_bins_1 <= 0)
            const double special = (n[b] + a[b]) / (1.0 + p_cal);
            A[b] = (u[b] == 0.0) ? special : r2;                    // np.choose(u_bins == 0, [A_bins_2, A_bins_special])
            if (!(0.0 <= A[b])) e->status |= BI_ST_BB_NEG;          // assert np.all(0 <= A_bins)
            aw[b] = A[b] * w[b];
        }
        const double tot = numpy_sum(aw.data(), B);
        for (int64_t b = 0; b < B; ++b) ps[(size_t)si * B + b] = aw[b] / tot;
        mus[si] = tot * p_cal;
    }

    // _compute_likelihood (:662-675)
    std::vector<double> term(B);
    for (int64_t b = 0; b < B; ++b) {
        double acc = 0.0;
        for (int s = 0; s < S; ++s) {
            const double x = ps[(size_t)s * B + b] * mus[s];
            acc = (s == 0) ? x : acc + x;
        }
        term[b] = poisson_logpmf(n[b], acc);
    }
    e->ll = numpy_sum(term.data(), B);
    if (keep) { e->mus = mus; e->ps = ps; }
}

int check_ready(bi_ctx* c, bool need_data) {
    if (!c) return BI_ERR_INVALID;
    if (!c->model_ready) return fail(c, BI_ERR_STATE, "no model uploaded");
    if (need_data && c->T == 0) return fail(c, BI_ERR_STATE, "no data uploaded");
    return BI_OK;
}

int declare_grid(bi_ctx* c, int d, const int32_t* n_anchor, const double* anchor_z, int S, int64_t B, int bb_source) {
    if (d < 0 || d > 16 || S < 1 || B < 0 || bb_source >= S || bb_source < -1) return fail(c, BI_ERR_INVALID, "bad model shape");
    if (d > 0 && (!n_anchor || !anchor_z)) return fail(c, BI_ERR_INVALID, "anchor grid is NULL");
    c->d = d; c->S = S; c->B = B; c->bb_source = bb_source;
    c->n_anchor.assign(d, 0);
    c->grid.assign(d, {});
    c->A = 1;
    const double* zp = anchor_z;
    for (int i = 0; i < d; ++i) {
        if (n_anchor[i] < 1) return fail(c, BI_ERR_INVALID, "axis %d has no anchors", i);
        c->n_anchor[i] = n_anchor[i];
        c->grid[i].assign(zp, zp + n_anchor[i]);
        for (int j = 1; j < n_anchor[i]; ++j)
            if (!(zp[j] > zp[j - 1])) return fail(c, BI_ERR_INVALID, "anchors of axis %d are not strictly ascending", i);
        zp += n_anchor[i];
        c->A *= n_anchor[i];
    }
    c->ps.assign((size_t)c->A * S * B, 0.0);
    c->mus.assign((size_t)c->A * S, 0.0);
    c->nm.assign(bb_source >= 0 ? (size_t)c->A * B : 0, 0.0);
    c->allow_negative.assign(S, 0);
    c->anchor_set.assign((size_t)c->A, 0);
    c->model_ready = false;
    c->T = 0;
    c->counts.clear();
    return BI_OK;
}

}  // namespace

extern "C" {

int bi_create(int device, bi_ctx** out) {
    (void)device;
    if (!out) return fail(nullptr, BI_ERR_INVALID, "out is NULL");
    *out = new bi_ctx();
    return BI_OK;
}

void bi_destroy(bi_ctx* ctx) { delete ctx; }

const char* bi_last_error(const bi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

const char* bi_version(void) { return "blueice_host 0.1 (reference-order C++ loops on the host; boundary test build, not the product)"; }

int bi_device_info(bi_ctx* ctx, char* name, char* arch, int len, int* n_cu, int64_t* hbm_bytes) {
    if (!ctx) return BI_ERR_INVALID;
    if (name && len > 0) snprintf(name, (size_t)len, "host");
    if (arch && len > 0) snprintf(arch, (size_t)len, "host");
    if (n_cu) *n_cu = 0;
    if (hbm_bytes) *hbm_bytes = 0;
    return BI_OK;
}

int bi_model_begin(bi_ctx* ctx, int d, const int32_t* n_anchor, const double* anchor_z, int S, int64_t B, int bb_source) {
    if (!ctx) return BI_ERR_INVALID;
    const int rc = declare_grid(ctx, d, n_anchor, anchor_z, S, B, bb_source);
    if (rc) return rc;
    ctx->streaming = true;
    return BI_OK;
}

int bi_model_set_anchor(bi_ctx* ctx, int64_t anchor_index, const double* ps, const double* mus, const double* n_model_row) {
    if (!ctx) return BI_ERR_INVALID;
    if (!ctx->streaming) return fail(ctx, BI_ERR_STATE, "bi_model_set_anchor outside bi_model_begin / bi_model_end");
    if (anchor_index < 0 || anchor_index >= ctx->A || !ps || !mus) return fail(ctx, BI_ERR_INVALID, "bad anchor index or NULL rows");
    if (ctx->bb_source >= 0 && !n_model_row) return fail(ctx, BI_ERR_INVALID, "Beeston-Barlow model without n_model row");
    const size_t SB = (size_t)ctx->S * ctx->B;
    std::memcpy(ctx->ps.data() + anchor_index * SB, ps, SB * sizeof(double));
    std::memcpy(ctx->mus.data() + anchor_index * ctx->S, mus, (size_t)ctx->S * sizeof(double));
    if (ctx->bb_source >= 0) std::memcpy(ctx->nm.data() + anchor_index * ctx->B, n_model_row, (size_t)ctx->B * sizeof(double));
    ctx->anchor_set[(size_t)anchor_index] = 1;
    return BI_OK;
}

int bi_model_end(bi_ctx* ctx) {
    if (!ctx) return BI_ERR_INVALID;
    if (!ctx->streaming) return fail(ctx, BI_ERR_STATE, "bi_model_end without bi_model_begin");
    for (int64_t a = 0; a < ctx->A; ++a)
        if (!ctx->anchor_set[(size_t)a]) return fail(ctx, BI_ERR_STATE, "anchor model %lld was never set", (long long)a);
    ctx->streaming = false;
    ctx->model_ready = true;
    return BI_OK;
}

int bi_upload_model(bi_ctx* ctx, int d, const int32_t* n_anchor, const double* anchor_z, int S, int64_t B, const double* ps,
                    const double* mus, const double* n_model, int bb_source) {
    if (!ctx) return BI_ERR_INVALID;
    if (!ps || !mus) return fail(ctx, BI_ERR_INVALID, "ps / mus is NULL");
    if (bb_source >= 0 && !n_model) return fail(ctx, BI_ERR_INVALID, "Beeston-Barlow model without n_model");
    int rc = bi_model_begin(ctx, d, n_anchor, anchor_z, S, B, bb_source);
    if (rc) return rc;
    const size_t SB = (size_t)S * B;
    for (int64_t a = 0; a < ctx->A; ++a) {
        rc = bi_model_set_anchor(ctx, a, ps + a * SB, mus + a * S, bb_source >= 0 ? n_model + a * SB + (size_t)bb_source * B : nullptr);
        if (rc) return rc;
    }
    return bi_model_end(ctx);
}

int bi_set_allow_negative(bi_ctx* ctx, const int32_t* allow) {
    if (!ctx || !allow) return BI_ERR_INVALID;
    if (ctx->S < 1) return fail(ctx, BI_ERR_STATE, "no model declared");
    for (int s = 0; s < ctx->S; ++s) ctx->allow_negative[s] = allow[s];
    return BI_OK;
}

int bi_upload_counts(bi_ctx* ctx, int64_t T, const double* counts) {
    const int rc = check_ready(ctx, false);
    if (rc) return rc;
    if (T < 1 || !counts) return fail(ctx, BI_ERR_INVALID, "bi_upload_counts: T < 1 or NULL");
    ctx->counts.assign(counts, counts + T * ctx->B);
    ctx->T = T;
    return BI_OK;
}

int bi_eval(bi_ctx* ctx, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, double* out, int32_t* status) {
    const int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (P < 0 || (P > 0 && !out) || (P > 0 && ctx->d > 0 && !z)) return fail(ctx, BI_ERR_INVALID, "bi_eval: bad arguments");
    Eval e;
    for (int64_t p = 0; p < P; ++p) {
        evaluate(ctx, z ? z + p * ctx->d : nullptr, rate_scale ? rate_scale + p * ctx->S : nullptr, dataset ? dataset[p] : 0, false, &e);
        out[p] = e.ll;
        if (status) status[p] = e.status;
    }
    return BI_OK;
}

int bi_eval_full(bi_ctx* ctx, const double* z, const double* rate_scale, int64_t dataset, double* ll, double* mus_out, double* ps_out,
                 int32_t* status) {
    const int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (ctx->d > 0 && !z) return fail(ctx, BI_ERR_INVALID, "bi_eval_full: z is NULL");
    Eval e;
    evaluate(ctx, z, rate_scale, dataset, true, &e);
    if (ll) *ll = e.ll;
    if (status) *status = e.status;
    if (!e.mus.empty()) {
        if (mus_out) std::memcpy(mus_out, e.mus.data(), e.mus.size() * sizeof(double));
        if (ps_out) std::memcpy(ps_out, e.ps.data(), e.ps.size() * sizeof(double));
    }
    return BI_OK;
}

int bi_interpolate(bi_ctx* ctx, int which, const double* z, double* out) {
    const int rc = check_ready(ctx, false);
    if (rc) return rc;
    if (!out || (ctx->d > 0 && !z)) return fail(ctx, BI_ERR_INVALID, "bi_interpolate: NULL argument");
    if (!in_box(ctx, z)) return fail(ctx, BI_ERR_INVALID, "One of the requested xi is out of bounds");
    const Cell cell = find_cell(ctx, z);
    if (which == 0) morph(cell, ctx->ps.data(), (int64_t)ctx->S * ctx->B, out);
    else if (which == 1) morph(cell, ctx->mus.data(), ctx->S, out);
    else if (which == 2 && ctx->bb_source >= 0) morph(cell, ctx->nm.data(), ctx->B, out);
    else return fail(ctx, BI_ERR_INVALID, "bi_interpolate: which = %d", which);
    return BI_OK;
}

int bi_set_param(bi_ctx* ctx, const char* name, int64_t value) {
    (void)value;
    return fail(ctx, BI_ERR_INVALID, "the host build has no parameter '%s'", name ? name : "(null)");
}

int64_t bi_get_param(bi_ctx* ctx, const char* name) {
    fail(ctx, BI_ERR_INVALID, "the host build has no parameter '%s'", name ? name : "(null)");
    return INT64_MIN;
}

int bi_list_params(char* buf, int len) {
    if (buf && len > 0) buf[0] = 0;
    return 1;
}

}  // extern "C"
