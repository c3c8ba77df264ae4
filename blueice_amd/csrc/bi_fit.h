// bi_fit.h -- the batched profile-fit engine's inner loop in C++ (host code; round 4).
//
// blueice_amd/profile.py advances P minimisations in lock-step, one bi_eval_grad call per optimiser iteration over every
// problem still running -- in place of the reference's loops of sequential scipy fits (blueice/inference.py:131-178 inside
// :332-443).  Its optimiser arithmetic was numpy on [P, F] arrays: ~60 small array operations per iteration, each paying
// numpy's per-call cost, which made the host two thirds of a profiled scan.  This is the same
// algorithm, statement by statement (BFGS with the Hessian estimate in direct form and exact reduced steps at bounds;
// kinks of the morph handled as kinks: both one-sided slopes for a variable sitting on an anchor, steps end at the first
// kink they would cross; Armijo backtracking with a ladder of three trial steps per call; crawl / rounding-floor
// detection), as loops over problems.
//   bi_minimize_batched   the optimiser over an objective callback  fun(x [n][F], rows [n]) -> f [n], g [n][F]
//   bi_fit_batched        ... with the device likelihood as the objective, no Python between the iterations: optimiser
//                         variable j is a shape parameter (z_i = x_j) or a rate multiplier (rate_scale_s = x_j * unit_s)
#pragma once

#include <new>
#include <stdexcept>

namespace {

struct FitState {
    int64_t P;
    int F;
    std::vector<double> x, f, g, B, reach, history;
    std::vector<char> fresh, done, failed, stalled, converged, crawled;
    std::vector<int32_t> flat;
};

// solve A d = b for one small dense system (partial pivoting); false if singular
bool solve_small(int n, double* A, double* b) {
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
        if (A[piv * n + c] == 0.0 || !(A[piv * n + c] == A[piv * n + c])) return false;
        if (piv != c) {
            for (int k = 0; k < n; ++k) std::swap(A[c * n + k], A[piv * n + k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < n; ++r) {
            const double m = A[r * n + c] / A[c * n + c];
            if (m == 0.0) continue;
            for (int k = c; k < n; ++k) A[r * n + k] -= m * A[c * n + k];
            b[r] -= m * b[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double v = b[r];
        for (int k = r + 1; k < n; ++k) v -= A[r * n + k] * b[k];
        b[r] = v / A[r * n + r];
    }
    return true;
}

inline double maxabs(const double* v, int n) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(v[i]));
    return m;
}

int minimize_batched(bi_objective_fn fun, void* user, int64_t P, int F, const double* x0, const double* lo, const double* hi,
                     const int32_t* n_kinks, const double* kinks_flat, double gtol, int max_iter, double* x_out, double* f_out,
                     int32_t* flags_out, int64_t* counters) {
    const double c1 = 1e-4, ftol = 1e-15, slow_tol = 1e-11, inf = std::numeric_limits<double>::infinity();
    const int max_halvings = 20, slow_window = 8;
    const double ladder[3] = {1.0, 0.25, 0.0625};
    std::vector<std::vector<double>> kinks((size_t)F);
    bool have_kinks = false;
    if (n_kinks) {
        const double* kp = kinks_flat;
        for (int j = 0; j < F; ++j) {
            kinks[(size_t)j].assign(kp, kp + n_kinks[j]);
            kp += n_kinks[j];
            have_kinks |= n_kinks[j] > 0;
        }
    }
    const size_t nP = (size_t)P, nF = (size_t)F;
    std::vector<double> x(nP * nF), f(nP), g(nP * nF), B(nP * nF * nF, 0.0), reach(nP, 1.0), history((size_t)slow_window * nP, inf);
    std::vector<char> fresh(nP, 1), done(nP, 0), failed(nP, 0), stalled(nP, 0), converged(nP, 0), crawled(nP, 0), blocked(nP * nF, 0);
    std::vector<int32_t> flat(nP, 0);
    std::vector<int64_t> rows(nP);
    std::vector<double> span(nF);
    for (int j = 0; j < F; ++j) span[(size_t)j] = std::isfinite(hi[j] - lo[j]) ? hi[j] - lo[j] : inf;
    for (int64_t p = 0; p < P; ++p) {
        rows[(size_t)p] = p;
        for (int j = 0; j < F; ++j) x[(size_t)p * nF + j] = std::min(std::max(x0[(size_t)p * nF + j], lo[j]), hi[j]);
        for (int j = 0; j < F; ++j) B[((size_t)p * nF + j) * nF + j] = 1.0;
    }
    auto set_identity = [&](int64_t p, double scale) {
        double* b = &B[(size_t)p * nF * nF];
        for (int a = 0; a < F; ++a)
            for (int c = 0; c < F; ++c) b[a * F + c] = a == c ? scale : 0.0;
    };
    int64_t calls = 0, kink_calls = 0;
    int rc = fun(user, P, F, x.data(), rows.data(), f.data(), g.data());
    if (rc) return rc;
    calls = 1;
    for (int64_t p = 0; p < P; ++p) done[(size_t)p] = failed[(size_t)p] = !std::isfinite(f[(size_t)p]);

    std::vector<double> xt, ft, gt, pg(nP * nF);
    std::vector<int64_t> rt;
    int it = 0;
    for (it = 1; it <= max_iter; ++it) {
        for (int64_t p = 0; p < P; ++p)
            for (int j = 0; j < F; ++j) {
                const size_t q = (size_t)p * nF + j;
                blocked[q] = (x[q] <= lo[j] && g[q] > 0) || (x[q] >= hi[j] && g[q] < 0);
            }
        if (have_kinks) {
            // variables sitting ON a kink: both one-sided slopes from two more rows, one ulp below and above
            std::vector<int64_t> kp;
            std::vector<int> kj;
            for (int64_t p = 0; p < P; ++p) {
                if (done[(size_t)p]) continue;
                for (int j = 0; j < F; ++j) {
                    const std::vector<double>& ks = kinks[(size_t)j];
                    if (!ks.empty() && std::binary_search(ks.begin(), ks.end(), x[(size_t)p * nF + j])) { kp.push_back(p); kj.push_back(j); }
                }
            }
            const size_t n = kp.size();
            if (n) {
                xt.assign(2 * n * nF, 0.0);
                rt.assign(2 * n, 0);
                for (size_t k = 0; k < n; ++k) {
                    const double* xp = &x[(size_t)kp[k] * nF];
                    std::copy(xp, xp + F, &xt[k * nF]);
                    std::copy(xp, xp + F, &xt[(n + k) * nF]);
                    xt[k * nF + kj[k]] = std::nextafter(xp[kj[k]], -inf);
                    xt[(n + k) * nF + kj[k]] = std::nextafter(xp[kj[k]], inf);
                    rt[k] = rt[n + k] = kp[k];
                }
                ft.assign(2 * n, 0.0);
                gt.assign(2 * n * nF, 0.0);
                if ((rc = fun(user, (int64_t)(2 * n), F, xt.data(), rt.data(), ft.data(), gt.data()))) return rc;
                ++calls;
                ++kink_calls;
                for (size_t k = 0; k < n; ++k) {
                    double gl = gt[k * nF + kj[k]], gr = gt[(n + k) * nF + kj[k]];
                    if (!(std::isfinite(ft[k]) && std::isfinite(gl))) gl = 0.0;
                    if (!(std::isfinite(ft[n + k]) && std::isfinite(gr))) gr = 0.0;
                    const bool hold = gr >= 0 && gl <= 0;              // uphill on both sides: a minimum along this variable
                    const bool go_left = !hold && gl > 0 && (gr >= 0 || gl > -gr);
                    const size_t q = (size_t)kp[k] * nF + kj[k];
                    g[q] = go_left ? gl : gr;
                    if (hold) blocked[q] = 1;
                }
                for (size_t k = 0; k < n; ++k) {                       // leaving a kink: a plain gradient step
                    const double gl0 = gt[k * nF + kj[k]], gr0 = gt[(n + k) * nF + kj[k]];
                    const double gl = (std::isfinite(ft[k]) && std::isfinite(gl0)) ? gl0 : 0.0;
                    const double gr = (std::isfinite(ft[n + k]) && std::isfinite(gr0)) ? gr0 : 0.0;
                    if (!(gr >= 0 && gl <= 0)) { set_identity(kp[k], 1.0); fresh[(size_t)kp[k]] = 1; }
                }
            }
        }
        const int slot = it % slow_window;
        int64_t n_act = 0;
        for (int64_t p = 0; p < P; ++p) {
            const size_t sp = (size_t)p;
            double m = 0.0;
            for (int j = 0; j < F; ++j) {
                const size_t q = sp * nF + j;
                pg[q] = blocked[q] ? 0.0 : g[q];
                m = std::max(m, std::fabs(pg[q]));                    // (nan compares false: it never raises the maximum)
            }
            bool nanpg = false;
            for (int j = 0; j < F; ++j) nanpg |= pg[sp * nF + j] != pg[sp * nF + j];
            if (!done[sp] && !nanpg && m <= gtol) { converged[sp] = 1; done[sp] = 1; }
            // hardly anything gained over the last slow_window iterations: the zigzag across a kink of the morph
            const double h = history[(size_t)slot * nP + sp];
            const bool crawling = !done[sp] && (h - f[sp] <= slow_tol * std::max(1.0, std::fabs(f[sp])));
            history[(size_t)slot * nP + sp] = f[sp];
            if (crawling && crawled[sp]) { stalled[sp] = 1; done[sp] = 1; }
            else if (crawling) {
                set_identity(p, 1.0);
                fresh[sp] = 1;
                for (int w = 0; w < slow_window; ++w) history[(size_t)w * nP + sp] = inf;
                crawled[sp] = 1;
            }
            if (!done[sp]) ++n_act;
        }
        if (!n_act) break;
        std::vector<int64_t> act;
        act.reserve((size_t)n_act);
        for (int64_t p = 0; p < P; ++p) if (!done[(size_t)p]) act.push_back(p);
        const size_t nA = act.size();
        std::vector<double> d(nA * nF), slope(nA), alpha(nA), clo(nA * nF), chi(nA * nF), xa(nA * nF), fa(nA), ga(nA * nF);
        std::vector<char> freev(nA * nF);
        std::vector<double> Bm(nF * nF), rhs(nF);
        for (size_t a = 0; a < nA; ++a) {
            const int64_t p = act[a];
            const size_t sp = (size_t)p;
            std::copy(&x[sp * nF], &x[sp * nF] + F, &xa[a * nF]);
            std::copy(&g[sp * nF], &g[sp * nF] + F, &ga[a * nF]);
            fa[a] = f[sp];
            for (int j = 0; j < F; ++j) freev[a * nF + j] = !blocked[sp * nF + j];
            // quasi-Newton step in the subspace of the free variables (rows / columns of the pinned ones replaced by identity)
            for (int r = 0; r < F; ++r)
                for (int c2 = 0; c2 < F; ++c2)
                    Bm[(size_t)r * nF + c2] = (freev[a * nF + r] && freev[a * nF + c2]) ? B[(sp * nF + r) * nF + c2] : (r == c2 ? 1.0 : 0.0);
            for (int j = 0; j < F; ++j) rhs[(size_t)j] = freev[a * nF + j] ? ga[a * nF + j] : 0.0;
            const bool solved = solve_small(F, Bm.data(), rhs.data());
            double sl = 0.0;
            bool fin = true;
            for (int j = 0; j < F; ++j) {
                const double dj = solved ? (freev[a * nF + j] ? -rhs[(size_t)j] : 0.0) : 0.0;
                d[a * nF + j] = dj;
                sl += dj * ga[a * nF + j];
                fin &= std::isfinite(dj);
            }
            if (!(sl < 0) || fresh[sp] || !fin) {
                // steepest descent; the step is ~1 long in x at the start, a few times the last accepted step later
                const double length = std::min(1.0, 4.0 * reach[sp]);
                double l1 = 0.0;
                for (int j = 0; j < F; ++j) l1 += std::fabs(pg[sp * nF + j]);
                const double den = std::max(1.0 / length, l1 / length);
                sl = 0.0;
                for (int j = 0; j < F; ++j) {
                    d[a * nF + j] = -pg[sp * nF + j] / den;
                    sl += d[a * nF + j] * ga[a * nF + j];
                }
                set_identity(p, 1.0);
                fresh[sp] = 1;
            }
            slope[a] = sl;
            double al = 1.0;
            for (int j = 0; j < F; ++j) {
                const double cap = (fresh[sp] ? 0.1 : 0.25) * span[(size_t)j] / std::max(std::fabs(d[a * nF + j]), 1e-300);
                if (cap < al) al = cap;
            }
            alpha[a] = al;
            // a step ends at the first kink it would cross: the box of this step is the grid cell it runs in
            for (int j = 0; j < F; ++j) {
                double l = lo[j], h2 = hi[j];
                const std::vector<double>& ks = kinks[(size_t)j];
                if (!ks.empty()) {
                    const double xv = xa[a * nF + j];
                    const auto up = std::upper_bound(ks.begin(), ks.end(), xv);
                    const auto dn = std::lower_bound(ks.begin(), ks.end(), xv);
                    const double nxt = up != ks.end() ? *up : hi[j];
                    const double prv = dn != ks.begin() ? *(dn - 1) : lo[j];
                    if (d[a * nF + j] > 0) h2 = std::min(nxt, hi[j]);
                    if (d[a * nF + j] < 0) l = std::max(prv, lo[j]);
                }
                clo[a * nF + j] = l;
                chi[a * nF + j] = h2;
            }
        }
        // Backtracking: the full step first; every later round a ladder of three steps per problem in the same call
        std::vector<size_t> todo(nA);
        for (size_t a = 0; a < nA; ++a) todo[a] = a;
        std::vector<double> acc_x(xa), acc_f(fa), acc_g(ga);
        std::vector<char> accepted(nA, 0);
        int spent = 0;
        while (spent < max_halvings && !todo.empty()) {
            const int K = spent == 0 ? 1 : 3;
            const size_t n = todo.size();
            xt.assign(n * K * nF, 0.0);
            rt.assign(n * K, 0);
            std::vector<double> al(n * K);
            for (size_t t = 0; t < n; ++t) {
                const size_t a = todo[t];
                for (int k = 0; k < K; ++k) {
                    al[t * K + k] = alpha[a] * ladder[k];
                    for (int j = 0; j < F; ++j) {
                        const double v = xa[a * nF + j] + al[t * K + k] * d[a * nF + j];
                        xt[(t * K + k) * nF + j] = std::min(std::max(v, clo[a * nF + j]), chi[a * nF + j]);
                    }
                    rt[t * K + k] = act[a];
                }
            }
            ft.assign(n * K, 0.0);
            gt.assign(n * K * nF, 0.0);
            if ((rc = fun(user, (int64_t)(n * K), F, xt.data(), rt.data(), ft.data(), gt.data()))) return rc;
            ++calls;
            spent += K;
            std::vector<size_t> miss;
            for (size_t t = 0; t < n; ++t) {
                const size_t a = todo[t];
                int pick = -1;
                for (int k = 0; k < K && pick < 0; ++k) {
                    const size_t r = t * K + k;
                    double lin = 0.0;
                    bool gfin = true;
                    for (int j = 0; j < F; ++j) {
                        lin += ga[a * nF + j] * (xt[r * nF + j] - xa[a * nF + j]);
                        gfin &= std::isfinite(gt[r * nF + j]);
                    }
                    if (std::isfinite(ft[r]) && ft[r] <= fa[a] + c1 * lin && gfin) pick = k;
                }
                if (pick >= 0) {
                    const size_t r = t * K + pick;
                    std::copy(&xt[r * nF], &xt[r * nF] + F, &acc_x[a * nF]);
                    std::copy(&gt[r * nF], &gt[r * nF] + F, &acc_g[a * nF]);
                    acc_f[a] = ft[r];
                    accepted[a] = 1;
                } else {
                    // quadratic interpolation from the shortest trial where it was finite, halving otherwise; kept in [0.1, 0.5] of it
                    const double a_m = al[t * K + K - 1], f_t = ft[t * K + K - 1];
                    const double quad = -slope[a] * a_m * a_m / (2.0 * (f_t - fa[a] - slope[a] * a_m));
                    alpha[a] = std::isfinite(quad) ? std::min(std::max(quad, 0.1 * a_m), 0.5 * a_m) : 0.5 * a_m;
                    miss.push_back(a);
                }
            }
            todo.swap(miss);
            if (todo.empty()) break;
            bool all_tiny = true;
            for (size_t a : todo)
                all_tiny &= alpha[a] * maxabs(&d[a * nF], F) < 1e-13 * std::max(1.0, maxabs(&xa[a * nF], F));
            if (all_tiny) break;
        }
        for (size_t a = 0; a < nA; ++a) {
            const int64_t p = act[a];
            const size_t sp = (size_t)p;
            if (!accepted[a]) {
                // no acceptable step: once more from a fresh B; if that was a fresh B already, this is where the problem can get
                if (!fresh[sp]) { set_identity(p, 1.0); fresh[sp] = 1; }
                else { stalled[sp] = 1; done[sp] = 1; }
                continue;
            }
            // BFGS update in the subspace of the variables that moved freely
            std::vector<double> s(nF), y(nF);
            double sy = 0.0, ss = 0.0, yy = 0.0, rch = 0.0;
            for (int j = 0; j < F; ++j) {
                const size_t q = a * nF + j;
                const bool pinned = !freev[q] || acc_x[q] <= clo[q] || acc_x[q] >= chi[q];
                s[(size_t)j] = pinned ? 0.0 : acc_x[q] - xa[q];
                y[(size_t)j] = pinned ? 0.0 : acc_g[q] - ga[q];
                sy += s[(size_t)j] * y[(size_t)j];
                ss += s[(size_t)j] * s[(size_t)j];
                yy += y[(size_t)j] * y[(size_t)j];
                rch = std::max(rch, std::fabs(acc_x[q] - xa[q]));
            }
            const bool good = sy > 1e-10 * std::sqrt(ss * yy);
            if (good) {
                if (fresh[sp]) set_identity(p, yy / sy);            // scale the first estimate (Nocedal & Wright 6.20)
                fresh[sp] = 0;
                double* b = &B[sp * nF * nF];
                std::vector<double> Bs(nF, 0.0);
                double sBs = 0.0;
                for (int r = 0; r < F; ++r) {
                    double v = 0.0;
                    for (int c2 = 0; c2 < F; ++c2) v += b[r * F + c2] * s[(size_t)c2];
                    Bs[(size_t)r] = v;
                }
                for (int r = 0; r < F; ++r) sBs += s[(size_t)r] * Bs[(size_t)r];
                for (int r = 0; r < F; ++r)
                    for (int c2 = 0; c2 < F; ++c2)
                        b[r * F + c2] = b[r * F + c2] - (Bs[(size_t)r] * Bs[(size_t)c2]) / sBs + (y[(size_t)r] * y[(size_t)c2]) / sy;
            }
            reach[sp] = std::max(rch, 1e-12);
            const double gain = f[sp] - acc_f[a];
            flat[sp] = gain <= ftol * std::max(1.0, std::fabs(acc_f[a])) ? flat[sp] + 1 : 0;
            std::copy(&acc_x[a * nF], &acc_x[a * nF] + F, &x[sp * nF]);
            std::copy(&acc_g[a * nF], &acc_g[a * nF] + F, &g[sp * nF]);
            f[sp] = acc_f[a];
            // two steps in a row without a measurable decrease: the rounding floor if B was fresh, else distrust B first
            if (flat[sp] >= 2) {
                if (fresh[sp]) { converged[sp] = 1; done[sp] = 1; }
                else { set_identity(p, 1.0); fresh[sp] = 1; flat[sp] = 0; }
            }
        }
    }
    if (it > max_iter) it = max_iter;
    std::copy(x.begin(), x.end(), x_out);
    std::copy(f.begin(), f.end(), f_out);
    for (int64_t p = 0; p < P; ++p)
        flags_out[p] = (converged[(size_t)p] ? 1 : 0) | (stalled[(size_t)p] ? 2 : 0) | (failed[(size_t)p] ? 4 : 0);
    if (counters) { counters[0] = it; counters[1] = calls; counters[2] = kink_calls; }
    return BI_OK;
}

// the device likelihood as the objective: f = -ll, g = -d ll / d x
struct DeviceObjective {
    bi_ctx* c;
    int F;
    const int32_t* var_kind;    // [F] 0: shape parameter (axis var_index), 1: rate multiplier (source var_index)
    const int32_t* var_index;
    const double* z0;           // [P][d]   the problems' shape settings (the floating ones are overwritten)
    const double* scale0;       // [P][S]   the problems' rate scales for FIXED multipliers
    const double* unit;         // [P][S]   d rate_scale / d multiplier (live time, efficiency)
    const int64_t* dataset;     // [P] or NULL
    std::vector<double> z, sc, ll, grad;
    std::vector<int64_t> ds;
    std::vector<int32_t> st;
    int64_t evaluations = 0;
};

int device_objective(void* user, int64_t n, int F, const double* x, const int64_t* rows, double* f, double* g) {
    DeviceObjective* o = (DeviceObjective*)user;
    bi_ctx* c = o->c;
    const int d = c->d, S = c->S;
    o->z.resize((size_t)n * std::max(d, 1));
    o->sc.resize((size_t)n * S);
    o->ll.resize((size_t)n);
    o->grad.resize((size_t)n * (d + S));
    o->st.resize((size_t)n);
    if (o->dataset) o->ds.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t p = rows[i];
        for (int k = 0; k < d; ++k) o->z[(size_t)i * d + k] = o->z0[p * d + k];
        for (int s = 0; s < S; ++s) o->sc[(size_t)i * S + s] = o->scale0[p * S + s];
        for (int j = 0; j < F; ++j) {
            if (o->var_kind[j] == 0) o->z[(size_t)i * d + o->var_index[j]] = x[i * F + j];
            else o->sc[(size_t)i * S + o->var_index[j]] = x[i * F + j] * o->unit[p * S + o->var_index[j]];
        }
        if (o->dataset) o->ds[(size_t)i] = o->dataset[p];
    }
    o->evaluations += n;
    const int rc = bi_eval_grad(c, n, d ? o->z.data() : nullptr, o->sc.data(), o->dataset ? o->ds.data() : nullptr, o->ll.data(),
                                o->grad.data(), o->st.data());
    if (rc) return rc;
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    for (int64_t i = 0; i < n; ++i) {
        const int64_t p = rows[i];
        const int32_t st = o->st[(size_t)i];
        if (st & BI_ST_INTERNAL) return fail(c, BI_ERR_HIP, "the device gave up waiting for a partial sum (in-launch reduction): GPU fault");
        double ll = o->ll[(size_t)i];
        if (st & (BI_ST_BB_ROOT1 | BI_ST_BB_NEG)) ll = qnan;          // a point to avoid, not an exception (bb_assert = 'nan')
        const bool bad = (st & (BI_ST_OUT_OF_BOUNDS | BI_ST_UNPHYSICAL)) != 0;
        f[i] = bad ? std::numeric_limits<double>::infinity() : -ll;
        for (int j = 0; j < F; ++j) {
            double v = o->var_kind[j] == 0 ? o->grad[(size_t)i * (d + S) + o->var_index[j]]
                                           : o->grad[(size_t)i * (d + S) + d + o->var_index[j]] * o->unit[p * S + o->var_index[j]];
            g[i * F + j] = bad ? qnan : -v;
        }
    }
    return BI_OK;
}

// the entry points' argument checks: at most 2^24 problems per call (the optimiser's state is ~(F^2 + 8 F) doubles per problem),
// kink counts >= 0 and every variable's kinks ascending (the line search finds the next kink by bisection)
constexpr int64_t kFitMaxProblems = (int64_t)1 << 24;
bool kinks_are_valid(int F, const int32_t* n_kinks, const double* kinks) {
    if (!n_kinks) return true;
    int64_t off = 0;
    for (int j = 0; j < F; ++j) {
        if (n_kinks[j] < 0 || (n_kinks[j] > 0 && !kinks)) return false;
        for (int q = 1; q < n_kinks[j]; ++q)
            if (!(kinks[off + q - 1] <= kinks[off + q])) return false;
        off += n_kinks[j];
    }
    return true;
}

}  // namespace

extern "C" {

int bi_minimize_batched(bi_objective_fn fun, void* user, int64_t P, int F, const double* x0, const double* lo, const double* hi,
                        const int32_t* n_kinks, const double* kinks, double gtol, int max_iter, double* x_out, double* f_out,
                        int32_t* flags_out, int64_t* counters) {
    if (!fun || P < 0 || P > kFitMaxProblems || F < 1 || F > 64 || !x0 || !lo || !hi || !x_out || !f_out || !flags_out) return BI_ERR_INVALID;
    if (!kinks_are_valid(F, n_kinks, kinks)) return BI_ERR_INVALID;
    if (P == 0) { if (counters) counters[0] = counters[1] = counters[2] = 0; return BI_OK; }
    // (nothing may unwind through the C ABI: the optimiser's state is std::vectors of P F^2 doubles)
    try {
        return minimize_batched(fun, user, P, F, x0, lo, hi, n_kinks, kinks, gtol, max_iter, x_out, f_out, flags_out, counters);
    } catch (const std::bad_alloc&) {
        return BI_ERR_NOMEM;
    } catch (const std::exception&) {
        return BI_ERR_INVALID;
    }
}

int bi_fit_batched(bi_ctx* c, int64_t P, int F, const int32_t* var_kind, const int32_t* var_index, const double* z0,
                   const double* scale0, const double* unit, const int64_t* dataset, const double* x0, const double* lo,
                   const double* hi, const int32_t* n_kinks, const double* kinks, double gtol, int max_iter, double* x_out,
                   double* f_out, int32_t* flags_out, int64_t* counters) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (P < 0 || P > kFitMaxProblems || F < 1 || F > 64 || !var_kind || !var_index || !scale0 || !unit || (c->d > 0 && !z0) || !x0 || !lo || !hi || !x_out ||
        !f_out || !flags_out)
        return fail(c, BI_ERR_INVALID, "bi_fit_batched: bad arguments");
    for (int j = 0; j < F; ++j)
        if ((var_kind[j] == 0 && (var_index[j] < 0 || var_index[j] >= c->d)) || (var_kind[j] == 1 && (var_index[j] < 0 || var_index[j] >= c->S)) ||
            (var_kind[j] != 0 && var_kind[j] != 1))
            return fail(c, BI_ERR_INVALID, "bi_fit_batched: variable %d is neither a shape parameter nor a rate multiplier of this model", j);
    if (!kinks_are_valid(F, n_kinks, kinks)) return fail(c, BI_ERR_INVALID, "bi_fit_batched: n_kinks must be >= 0 and the kinks of a variable ascending");
    if (P == 0) { if (counters) counters[0] = counters[1] = counters[2] = counters[3] = 0; return BI_OK; }
    DeviceObjective o{};
    o.c = c; o.F = F; o.var_kind = var_kind; o.var_index = var_index; o.z0 = z0; o.scale0 = scale0; o.unit = unit; o.dataset = dataset;
    try {
        rc = minimize_batched(device_objective, &o, P, F, x0, lo, hi, n_kinks, kinks, gtol, max_iter, x_out, f_out, flags_out, counters);
    } catch (const std::bad_alloc&) {
        return fail(c, BI_ERR_NOMEM, "bi_fit_batched: out of host memory for %lld problems of %d variables", (long long)P, F);
    } catch (const std::exception& e) {
        return fail(c, BI_ERR_INVALID, "bi_fit_batched: %s", e.what());
    }
    if (counters) counters[3] = o.evaluations;
    return rc;
}

}  // extern "C"
