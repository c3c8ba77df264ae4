// bi_geometry.h -- host-side scalar half of one evaluation: grid cell and
// corner weights with scipy's RegularGridInterpolator semantics, rates, the reference's early exits.
#pragma once

namespace {

// scipy find_indices semantics on one axis (oracle/blueice_oracle.py:find_cell)
inline void find_cell(const std::vector<double>& g, double z, int& k, double& t) {
    const int n = (int)g.size();
    if (n == 1) { k = 0; t = 0.0; return; }
    if (z == g[n - 1]) {
        k = n - 2;
    } else {
        k = (int)(std::upper_bound(g.begin(), g.end(), z) - g.begin()) - 1;
        k = std::min(std::max(k, 0), n - 2);
    }
    const double denom = g[k + 1] - g[k];
    t = (z - g[k]) / denom;
}

struct PointGeom {
    int64_t cell_anchor;          // linear anchor index of the lower corner
    std::vector<double> w;        // [2^deff] corner weights, reference order
    double t[kMaxDim];            // per axis: normalised distance in the cell
    double inv_delta[kMaxDim];    // per axis: 1 / (g[k+1] - g[k])  (0 for single-anchor axes)
};

// corner c (bit i from the most significant = effective axis 0) -> anchor offset
inline int64_t corner_offset(const bi_ctx* c, int corner) {
    const int de = (int)c->eff_axes.size();
    int64_t off = 0;
    for (int i = 0; i < de; ++i)
        if ((corner >> (de - 1 - i)) & 1) off += c->astride[c->eff_axes[i]];
    return off;
}

// returns false when z is outside the anchor box (or nan): likelihood.py:345-347
bool point_geometry(const bi_ctx* c, const double* z, PointGeom& g) {
    for (int i = 0; i < c->d; ++i) {
        const auto& gr = c->grid[i];
        if (!(gr.front() <= z[i] && z[i] <= gr.back())) return false;
    }
    const int de = (int)c->eff_axes.size();
    int kk[kMaxDim];
    double tt[kMaxDim];
    int64_t base = 0;
    for (int i = 0; i < c->d; ++i) {
        int k; double t;
        find_cell(c->grid[i], z[i], k, t);
        base += (int64_t)k * c->astride[i];
        kk[i] = k; tt[i] = t;
        g.t[i] = t;
        g.inv_delta[i] = c->grid[i].size() > 1 ? 1.0 / (c->grid[i][(size_t)k + 1] - c->grid[i][(size_t)k]) : 0.0;
    }
    (void)kk;
    g.cell_anchor = base;
    const int nc = 1 << de;
    g.w.assign(nc, 1.0);
    for (int corner = 0; corner < nc; ++corner) {
        double w = 1.0;
        for (int i = 0; i < de; ++i) {
            const double t = tt[c->eff_axes[i]];
            const double wi = ((corner >> (de - 1 - i)) & 1) ? t : (1 - t);
            w = w * wi;
        }
        g.w[corner] = w;
    }
    return true;
}

// mus_interpolator(z): value = value + V*w per corner, left to right from 0.0
void interp_mus(const bi_ctx* c, const PointGeom& g, double* mus) {
    const int nc = (int)g.w.size();
    for (int s = 0; s < c->S; ++s) {
        double v = 0.0;
        for (int corner = 0; corner < nc; ++corner) {
            const int64_t a = g.cell_anchor + corner_offset(c, corner);
            const double term = c->h_mus[a * c->S + s] * g.w[corner];
            v = v + term;
        }
        mus[s] = v;
    }
}

// likelihood.py:397-415
bool rates_physical(const bi_ctx* c, const double* mus) {
    const double inf = std::numeric_limits<double>::infinity();
    bool any_allowed = false;
    for (int s = 0; s < c->S; ++s) any_allowed |= (c->allow_neg[s] != 0);
    if (!any_allowed) {
        for (int s = 0; s < c->S; ++s)
            if (!(mus[s] >= 0 && mus[s] < inf)) return false;
        return true;
    }
    bool any_fin = false;
    double tot = 0;
    for (int s = 0; s < c->S; ++s) { any_fin |= (mus[s] < inf); tot += mus[s]; }
    if (!any_fin || tot < 0) return false;
    for (int s = 0; s < c->S; ++s)
        if (!(0 <= mus[s]) && !c->allow_neg[s]) return false;
    return true;
}

int pick_class(int n, int maxg) {
    int g = 1;
    while (g < n && g < maxg) g <<= 1;
    return g;
}

}  // namespace
