"""Batched profile fits: P independent maximisations of one likelihood advanced in lock-step on the device.

The reference profiles a likelihood by running `bestfit_routine(lf, **fixed)` once per hypothesis, one scalar `lf()` call
after the other -- the loops of `plot_likelihood_ratio` (blueice/inference.py:424-432) and the brentq search of
`one_parameter_interval` (:332-389, 3 387 calls per limit in SURVEY.md's probe).  Here the P hypotheses are P
problems that differ only in their fixed parameters; every iteration of the optimiser is ONE device call that returns
value and analytic gradient for all problems still running (`lf.values_and_gradients` -> `bi_eval_grad`), and the
optimiser's own arithmetic is a few numpy operations on [P, F] arrays (F = floating parameters, a handful).

Optimiser: BFGS per problem (dense [F, F] Hessian estimates, updated together with einsum, solved together), Armijo
backtracking -- one more device call per halving, over the problems that still need it --, box constraints by
projection (rate multipliers >= 0 unless the source may go negative, shape parameters inside their anchor range;
variables pinned at a bound leave the search direction).  Likelihoods without an analytic gradient (unbinned; sums)
get central differences from batched `eval_points` calls instead.

    best, ll = bestfit_batched(lf, points={'shift': grid}, s2_rate_multiplier=1.)     # arrays [len(grid)]
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from .exceptions import DeviceError, NoOpimizationNecessary
from .utils import is_numeric

# The optimiser's inner loop: 'native' = the C++ port inside libblueice_hip.so (bi_minimize_batched / bi_fit_batched,
# csrc/bi_fit.h: the same algorithm as loops over problems, and with a device likelihood as the objective no Python between
# the iterations at all), 'numpy' = the array form below (kept as the executable specification: tests hold one against the other)
ENGINE = os.environ.get('BLUEICE_AMD_ENGINE', 'native')

__all__ = ['bestfit_batched', 'batched_minimize', 'batched_minimize_numpy', 'BatchObjective', 'supports_batched_fits']


def supports_batched_fits(lf):
    return hasattr(lf, 'eval_points') and hasattr(lf, 'get_bounds')


class BatchObjective:
    """f(x [n, F], rows [n]) -> (-ll [n], d(-ll)/dx [n, F]) for problems `rows` of P, whose fixed parameters are
    `points` (name -> array [P]) and `fixed` (name -> scalar)."""

    def __init__(self, lf, float_names, points, fixed, livetime_days=None, datasets=None):
        self.lf, self.names = lf, list(float_names)
        self.points = {k: np.asarray(v, dtype=float) for k, v in points.items()}
        self.fixed = dict(fixed)
        self.livetime_days = livetime_days
        self.datasets = None if datasets is None else np.asarray(datasets, dtype=np.int64)   # [P]: the dataset of every problem
        self.kinks = None                                     # per floating parameter: where differences must not straddle
        self.analytic = bool(getattr(lf, 'supports_gradient', False)) and hasattr(lf, 'values_and_gradients')
        # does the likelihood take `bb_assert` (a trial point at which a Beeston-Barlow assertion of the reference would fire
        # is a point to avoid, not an exception)?  Decided once, from the signature: a TypeError raised INSIDE a call is an error
        self.bb_assert = {}
        if self.analytic:
            import inspect
            try:
                params = inspect.signature(lf.values_and_gradients).parameters
                if 'bb_assert' in params or any(p.kind == p.VAR_KEYWORD for p in params.values()):
                    self.bb_assert = {'bb_assert': 'nan'}
            except (TypeError, ValueError):
                pass
        self.calls = self.evaluations = 0

    def native(self):
        """What bi_fit_batched needs to evaluate this objective without Python: the problems' settings as arrays, and which
        optimiser variable is which parameter -- or None where a Python callable sits between x and the device call (priors,
        a shape parameter that doubles as an efficiency, 'unphysical_behaviour': 'error', likelihoods that are not one
        device context)."""
        if hasattr(self, '_native'):
            return self._native
        self._native = None
        lf = self.lf
        ok = self.analytic and hasattr(lf, '_batch_terms') and getattr(lf, 'ctx', None) is not None and \
            hasattr(lf.ctx, 'fit_batched') and not any(getattr(lf, 'source_apply_efficiency', [])) and \
            lf.config.get('unphysical_behaviour') != 'error' and \
            all(v[1] is None for v in lf.shape_parameters.values()) and all(v is None for v in lf.rate_parameters.values())
        if not ok:
            return None
        P = max([len(v) for v in self.points.values()] + ([len(self.datasets)] if self.datasets is not None else []) + [1])
        pts = {k: np.broadcast_to(v, (P,)) for k, v in self.points.items()}
        pts.update(self.fixed)
        z0, scale0, prior, unit = lf._batch_terms(pts, self.livetime_days, want_unit=True)
        z0, scale0, unit = (np.ascontiguousarray(np.broadcast_to(a, (P,) + a.shape[1:]), dtype=float) for a in (z0, scale0, unit))
        kind, index = [], []
        shape_names = list(lf.shape_parameters)
        for n in self.names:
            if n in lf.shape_parameters:
                kind.append(0)
                index.append(shape_names.index(n))
            elif n.endswith('_rate_multiplier') and n[:-16] in lf.source_name_list:
                kind.append(1)
                index.append(lf.source_name_list.index(n[:-16]))
            else:
                return None
        self._native = dict(P=P, z0=z0, scale0=scale0, unit=unit, kind=np.array(kind, dtype=np.int32), index=np.array(index, dtype=np.int32))
        return self._native

    def stacked(self, period):
        """The same objective over several copies of the problem set: row r is problem r % period (multi-start fits)."""
        return StackedObjective(self, period)

    def _points_of(self, x, rows):
        pts = {k: v[rows] for k, v in self.points.items()}
        pts.update(self.fixed)
        for j, n in enumerate(self.names):
            pts[n] = x[:, j]
        return pts

    def __call__(self, x, rows):
        self.calls += 1
        more = {} if self.datasets is None else {'dataset': self.datasets[rows]}
        if self.analytic:
            self.evaluations += len(x)
            ll, grads = self.lf.values_and_gradients(self._points_of(x, rows), livetime_days=self.livetime_days, **self.bb_assert, **more)
            g = np.stack([np.broadcast_to(grads[n], ll.shape) for n in self.names], axis=1)
            return -ll, -g
        # central differences, all 2F + 1 stencil points of all problems in one batched call
        n, F = x.shape
        h = 1e-6 * np.maximum(1.0, np.abs(x))
        stencil = np.repeat(x[None], 2 * F + 1, axis=0)               # [2F+1, n, F]
        for j in range(F):
            stencil[1 + 2 * j, :, j] += h[:, j]
            stencil[2 + 2 * j, :, j] -= h[:, j]
        pts = self._points_of(stencil.reshape(-1, F), np.tile(rows, 2 * F + 1))
        self.evaluations += len(stencil) * n
        if more:
            more = {'dataset': np.tile(more['dataset'], 2 * F + 1)}
        ll = np.asarray(self.lf.eval_points(pts, livetime_days=self.livetime_days, **more)).reshape(2 * F + 1, n)
        g = np.empty((n, F))
        for j in range(F):
            up, dn = ll[1 + 2 * j], ll[2 + 2 * j]
            can_up, can_dn = np.isfinite(up), np.isfinite(dn)
            # a difference must not straddle a kink of the morph: ON one the forward difference (the right slope, which is
            # what the analytic gradient reports there), next to one the difference that stays on this side of it
            ks = self.kinks[j] if self.kinks is not None else ()
            if len(ks):
                xj, hj = x[:, j], h[:, j]
                on = np.isin(xj, ks)
                above = np.searchsorted(ks, xj, side='right')          # first kink > x
                below = np.searchsorted(ks, xj, side='left') - 1       # last kink < x
                kink_up = (above < len(ks)) & (ks[np.minimum(above, len(ks) - 1)] <= xj + hj)
                kink_dn = (below >= 0) & (ks[np.maximum(below, 0)] >= xj - hj)
                can_dn = can_dn & ~on & ~kink_dn
                can_up = can_up & ~(kink_up & ~on)
            with np.errstate(all='ignore'):                            # at a bound: the one-sided difference
                g[:, j] = np.where(can_up & can_dn, (up - dn) / (2 * h[:, j]),
                                   np.where(can_up, (up - ll[0]) / h[:, j], (ll[0] - dn) / h[:, j]))
        return -ll[0], -g


class StackedObjective:
    def __init__(self, base, period):
        self.base, self.period = base, int(period)

    def __call__(self, x, rows):
        return self.base(x, rows % self.period)


def _native_minimize(fun, x0, lo, hi, gtol, max_iter, kinks):
    """batched_minimize on the C++ loop of libblueice_hip.so.  A BatchObjective (or a stack of one) over a device likelihood
    without Python-side terms runs entirely inside bi_fit_batched; any other objective is called back from bi_minimize_batched."""
    from . import _capi
    lib = _capi.load()
    x0 = np.ascontiguousarray(x0, dtype=float)
    P, F = x0.shape
    lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lo, dtype=float), (F,)))
    hi = np.ascontiguousarray(np.broadcast_to(np.asarray(hi, dtype=float), (F,)))
    if kinks is not None and any(len(k) for k in kinks):
        n_k = np.array([len(k) for k in kinks], dtype=np.int32)
        k_flat = np.ascontiguousarray(np.concatenate([np.sort(np.asarray(k, dtype=float)) for k in kinks]))
    else:
        n_k = k_flat = None
    x = np.empty((P, F))
    f = np.empty(P)
    flags = np.zeros(P, dtype=np.int32)
    counters = np.zeros(4, dtype=np.int64)
    ptr = _capi.ptr
    base = fun.base if isinstance(fun, StackedObjective) else fun
    nat = base.native() if isinstance(base, BatchObjective) else None
    if nat is not None and (isinstance(fun, StackedObjective) or P == nat['P']):
        reps = P // nat['P'] if isinstance(fun, StackedObjective) else 1
        tile = (lambda a: np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1)))) if reps > 1 else (lambda a: a)
        ds = None if base.datasets is None else np.ascontiguousarray(tile(np.broadcast_to(base.datasets, (nat['P'],))), dtype=np.int64)
        z0, s0, un = tile(nat['z0']), tile(nat['scale0']), tile(nat['unit'])
        rc = base.lf.ctx.fit_batched(P, F, nat['kind'], nat['index'], z0, s0, un, ds, x0, lo, hi, n_k, k_flat, gtol, max_iter, x, f, flags, counters)
        base.calls += int(counters[1])
        base.evaluations += int(counters[3])
    else:
        err = []

        def callback(user, n, nf, xp, rp, fp, gp):
            try:
                xx = np.ctypeslib.as_array(xp, (n, nf)).copy()
                rr = np.ctypeslib.as_array(rp, (n,)).copy()
                ff, gg = fun(xx, rr)
                np.ctypeslib.as_array(fp, (n,))[:] = ff
                np.ctypeslib.as_array(gp, (n, nf))[:] = gg
                return 0
            except BaseException as e:          # (must not propagate through the C frames)
                err.append(e)
                return -1

        cb = _capi.OBJECTIVE_FN(callback)
        rc = lib.bi_minimize_batched(C.cast(cb, C.c_void_p), None, P, F, ptr(x0), ptr(lo), ptr(hi), ptr(n_k), ptr(k_flat), float(gtol),
                                     int(max_iter), ptr(x), ptr(f), ptr(flags), ptr(counters))
        if err:
            raise err[0]
        if rc:
            raise DeviceError("bi_minimize_batched failed (%d)" % rc)
    info = dict(converged=(flags & 1) != 0, stalled=(flags & 2) != 0, failed=(flags & 4) != 0, iterations=int(counters[0]),
                calls=int(counters[1]), kink_calls=int(counters[2]))
    return x, f, info


def batched_minimize(fun, x0, lo, hi, gtol=1e-6, max_iter=200, kinks=None, engine=None, **options):
    """Minimise P independent functions of F variables each (see `batched_minimize_numpy` for the algorithm): on the C++ loop
    of the library unless engine / BLUEICE_AMD_ENGINE says 'numpy' or non-default tuning options are given."""
    engine = engine or ENGINE
    if engine == 'native' and not options and np.ndim(x0) == 2 and np.shape(x0)[0] > 0 and 1 <= np.shape(x0)[1] <= 64:
        return _native_minimize(fun, x0, lo, hi, gtol, max_iter, kinks)
    return batched_minimize_numpy(fun, x0, lo, hi, gtol=gtol, max_iter=max_iter, kinks=kinks, **options)


def batched_minimize_numpy(fun, x0, lo, hi, gtol=1e-6, max_iter=200, c1=1e-4, max_halvings=20, ftol=1e-15, slow_window=8, slow_tol=1e-11,
                           kinks=None):
    """Minimise P independent functions of F variables each.  fun(x [n, F], rows [n]) -> (f [n], g [n, F]).
    lo / hi [F]: box (+-inf = none).  kinks: per variable, the interior points at which f has a kink along it (the anchors
    of a shape parameter: the morph is linear between them).  A variable sitting ON one -- every fit starts there, the
    base value of a shape parameter is an anchor -- has two one-sided slopes, taken from two more rows evaluated just
    below and just above it (whichever of the two `fun` reports AT the kink is its own convention).  Both pointing
    uphill: the variable is held like one at a bound (and the point can count as converged); otherwise the search
    continues down the steeper side.  -> (x [P, F], f [P], info) with info['converged'] [P] (projected gradient below
    gtol, or two successive steps that lowered f by less than ftol * max(1, |f|): the rounding floor),
    info['stalled'] [P] (no descent step found, or less than slow_tol * max(1, |f|) gained over the last slow_window
    iterations: a kink of the morph, where the search zigzags between two grid cells), info['iterations'], info['calls']."""
    x = np.clip(np.array(x0, dtype=float), lo, hi)
    P, F = x.shape
    rows_all = np.arange(P)
    f, g = fun(x, rows_all)
    calls = 1
    eye = np.eye(F)
    B = np.broadcast_to(eye, (P, F, F)).copy()          # Hessian estimates (direct BFGS form: a pinned variable can be cut out exactly)
    fresh = np.ones(P, dtype=bool)                       # B is still the (unscaled) identity
    done = ~np.isfinite(f)                               # nothing to descend from
    failed = done.copy()
    stalled = np.zeros(P, dtype=bool)
    converged = np.zeros(P, dtype=bool)
    flat = np.zeros(P, dtype=np.int32)                   # successive steps without a measurable decrease
    history = np.full((slow_window, P), np.inf)          # f of the last slow_window iterations (ring buffer)
    crawled = np.zeros(P, dtype=bool)
    reach = np.ones(P)                                   # length (max norm) of the last accepted step: the scale of a fresh gradient step
    at_lo = lambda xx: xx <= lo
    at_hi = lambda xx: xx >= hi
    # no step carries a boxed variable (a shape parameter) further than a quarter of its range -- a tenth on a plain
    # gradient step --: the morph is only piecewise smooth, and a long first stride lands in another grid cell's basin
    span = np.where(np.isfinite(hi - lo), hi - lo, np.inf)
    kinks = None if kinks is None or not any(len(k) for k in kinks) else [np.asarray(k, dtype=float) for k in kinks]
    kink_calls = 0
    it = 0
    for it in range(1, max_iter + 1):
        blocked = (at_lo(x) & (g > 0)) | (at_hi(x) & (g < 0))       # moving against the gradient would leave the box
        if kinks is not None:
            live = np.flatnonzero(~done)
            on = np.zeros((len(live), F), dtype=bool)
            for j, ks in enumerate(kinks):
                if len(ks):
                    on[:, j] = np.isin(x[live, j], ks)
            pi, ji = np.nonzero(on)
            if len(pi):
                rows_k = live[pi]
                k = np.arange(len(pi))
                xt = np.concatenate([x[rows_k], x[rows_k]])          # one row just below, one just above the kink
                xt[k, ji] = np.nextafter(xt[k, ji], -np.inf)
                xt[len(pi) + k, ji] = np.nextafter(xt[len(pi) + k, ji], np.inf)
                ft, gt = fun(xt, np.concatenate([rows_k, rows_k]))
                calls += 1
                kink_calls += 1
                gl, gr = gt[k, ji], gt[len(pi) + k, ji]
                gl = np.where(np.isfinite(ft[:len(pi)]) & np.isfinite(gl), gl, 0.0)
                gr = np.where(np.isfinite(ft[len(pi):]) & np.isfinite(gr), gr, 0.0)
                hold = (gr >= 0) & (gl <= 0)                         # uphill on both sides: a minimum along this variable
                go_left = ~hold & (gl > 0) & ((gr >= 0) | (gl > -gr))
                g[rows_k, ji] = np.where(go_left, gl, gr)            # the slope of the side the search goes down
                blocked[rows_k[hold], ji[hold]] = True
                # leaving a kink: a plain gradient step, whose components have the signs of the sides just chosen
                away = np.unique(rows_k[~hold])
                B[away] = eye
                fresh[away] = True
        pg = np.where(blocked, 0.0, g)
        converged |= ~done & (np.max(np.abs(pg), axis=1, initial=0.0) <= gtol)
        done |= converged
        # hardly anything gained over the last slow_window iterations: the zigzag across a kink of the morph
        slot = it % slow_window
        with np.errstate(invalid='ignore'):
            crawling = ~done & (history[slot] - f <= slow_tol * np.maximum(1.0, np.abs(f)))
        history[slot] = f
        # ... the first time, distrust the Hessian estimate and go on with gradient steps; the second time, stop
        again = crawling & crawled
        first = crawling & ~crawled
        if np.any(first):
            B[first] = eye
            fresh[first] = True
            history[:, first] = np.inf
            crawled[first] = True
        stalled |= again
        done |= again
        act = np.flatnonzero(~done)
        if not len(act):
            break
        ga, free = g[act], ~blocked[act]
        # quasi-Newton step in the subspace of the free variables: rows / columns of the pinned ones replaced by identity
        # (the sub-block of the Hessian estimate is inverted -- the sub-block of an inverse estimate would not be the same)
        pair = free[:, :, None] & free[:, None, :]
        Bm = np.where(pair, B[act], eye)
        try:
            d = -np.linalg.solve(Bm, (ga * free)[:, :, None])[:, :, 0] * free
        except np.linalg.LinAlgError:
            d = np.zeros_like(ga)
        slope = np.sum(d * ga, axis=1)
        reset = ~(slope < 0) | fresh[act] | ~np.all(np.isfinite(d), axis=1)
        if np.any(reset):      # steepest descent; the step is ~1 long in x at the start, a few times the last accepted step later
            pga = pg[act][reset]   # (near the optimum a unit step would cost twenty halvings, each a device call)
            length = np.minimum(1.0, 4.0 * reach[act][reset])[:, None]
            d[reset] = -pga / np.maximum(1.0 / length, np.sum(np.abs(pga), axis=1, keepdims=True) / length)
            B[act[reset]] = eye
            fresh[act[reset]] = True
            slope = np.sum(d * ga, axis=1)
        with np.errstate(all='ignore'):
            cap = np.where(fresh[act], 0.1, 0.25)[:, None] * span[None, :] / np.maximum(np.abs(d), 1e-300)
        alpha = np.minimum(1.0, np.min(cap, axis=1))
        xa, fa = x[act], f[act]
        # a step ends at the first kink it would cross (the function is smooth only up to there): the box of this step
        # is the grid cell the step runs in; the next iteration decides AT the kink whether and how to go on
        clo, chi = np.broadcast_to(lo, xa.shape), np.broadcast_to(hi, xa.shape)
        if kinks is not None:
            clo, chi = clo.copy(), chi.copy()
            for j, ks in enumerate(kinks):
                if not len(ks):
                    continue
                up = np.searchsorted(ks, xa[:, j], side='right')
                dn = np.searchsorted(ks, xa[:, j], side='left') - 1
                nxt = np.where(up < len(ks), ks[np.minimum(up, len(ks) - 1)], hi[j])
                prv = np.where(dn >= 0, ks[np.maximum(dn, 0)], lo[j])
                chi[:, j] = np.where(d[:, j] > 0, np.minimum(nxt, hi[j]), hi[j])
                clo[:, j] = np.where(d[:, j] < 0, np.maximum(prv, lo[j]), lo[j])
        todo = np.arange(len(act))
        acc_x, acc_f, acc_g = xa.copy(), fa.copy(), ga.copy()
        accepted = np.zeros(len(act), dtype=bool)
        # Backtracking.  The first trial is the full step (most problems take it).  Every later round tries a LADDER of three
        # steps per problem in the same device call -- the interpolated one, a quarter and a sixteenth of it -- and takes the
        # longest that satisfies the Armijo condition: a device call costs the same for three times the rows, and the number
        # of calls per iteration is set by the slowest problem.
        ladder = np.array([1.0, 0.25, 0.0625])
        spent = 0
        while spent < max_halvings:
            K = 1 if spent == 0 else len(ladder)
            al = alpha[todo][:, None] * ladder[None, :K]                                        # [n, K]
            xt = np.clip(xa[todo][:, None, :] + al[:, :, None] * d[todo][:, None, :], clo[todo][:, None, :], chi[todo][:, None, :])
            ft, gt = fun(xt.reshape(-1, F), np.repeat(act[todo], K))
            calls += 1
            spent += K
            ft, gt = ft.reshape(-1, K), gt.reshape(-1, K, F)
            with np.errstate(invalid='ignore'):
                ok = np.isfinite(ft) & (ft <= fa[todo][:, None] + c1 * np.sum(ga[todo][:, None, :] * (xt - xa[todo][:, None, :]), axis=2)) & \
                     np.all(np.isfinite(gt), axis=2)
            some = ok.any(axis=1)
            pick = np.argmax(ok, axis=1)[some]                       # the longest acceptable step of the ladder
            hit = todo[some]
            acc_x[hit], acc_f[hit], acc_g[hit] = xt[some, pick], ft[some, pick], gt[some, pick]
            accepted[hit] = True
            # quadratic interpolation from the shortest trial where it was finite, halving otherwise; kept in [0.1, 0.5] of it
            miss = todo[~some]
            if not len(miss):
                break
            a_m, f_t = al[~some, K - 1], ft[~some, K - 1]
            with np.errstate(all='ignore'):
                quad = -slope[miss] * a_m ** 2 / (2.0 * (f_t - fa[miss] - slope[miss] * a_m))
            alpha[miss] = np.where(np.isfinite(quad), np.clip(quad, 0.1 * a_m, 0.5 * a_m), 0.5 * a_m)
            todo = miss
            if np.all(alpha[todo] * np.max(np.abs(d[todo]), axis=1) < 1e-13 * np.maximum(1.0, np.max(np.abs(xa[todo]), axis=1))):
                break
        # problems without an acceptable step: once more from a fresh B; if that was a fresh B already, they are where
        # they can get (a kink between two grid cells of the morph, or the rounding floor of the likelihood)
        lost = ~accepted
        again = lost & ~fresh[act]
        B[act[again]] = eye
        fresh[act[again]] = True
        gone = lost & ~again
        stalled[act[gone]] = True
        done[act[gone]] = True
        # BFGS update of the others
        w = np.flatnonzero(accepted)
        if len(w):
            s = acc_x[w] - xa[w]
            y = acc_g[w] - ga[w]
            # the update lives in the subspace of the variables that moved freely: one that was pinned, or that the step
            # ran into a bound with (a clipped, arbitrarily short move against a finite change of slope), would plant a
            # huge curvature in B and shrink every later step to nothing
            pinned = ~free[w] | (acc_x[w] <= clo[w]) | (acc_x[w] >= chi[w])
            s = np.where(pinned, 0.0, s)
            y = np.where(pinned, 0.0, y)
            sy = np.sum(s * y, axis=1)
            good = sy > 1e-10 * np.sqrt(np.sum(s * s, axis=1) * np.sum(y * y, axis=1))
            rows = act[w]
            first = fresh[rows] & good
            if np.any(first):                                        # scale the first estimate (Nocedal & Wright 6.20)
                yy = np.sum(y[first] * y[first], axis=1)
                B[rows[first]] = eye * (yy / sy[first])[:, None, None]
            fresh[rows[good]] = False
            if np.any(good):
                sg, yg = s[good], y[good]
                Bg = B[rows[good]]
                Bs = np.einsum('pij,pj->pi', Bg, sg)
                sBs = np.sum(sg * Bs, axis=1)
                Bg = Bg - (Bs[:, :, None] * Bs[:, None, :]) / sBs[:, None, None] + (yg[:, :, None] * yg[:, None, :]) / sy[good][:, None, None]
                B[rows[good]] = Bg
            reach[rows] = np.maximum(np.max(np.abs(acc_x[w] - xa[w]), axis=1), 1e-12)
            gain = f[rows] - acc_f[w]
            flat[rows] = np.where(gain <= ftol * np.maximum(1.0, np.abs(acc_f[w])), flat[rows] + 1, 0)
            x[rows], f[rows], g[rows] = acc_x[w], acc_f[w], acc_g[w]
            # two steps in a row without a measurable decrease: if B was fresh (plain gradient steps) this is the rounding
            # floor of the function; otherwise distrust B first and take gradient steps from here
            floor = flat[rows] >= 2
            at_floor = rows[floor & fresh[rows]]
            converged[at_floor] = True
            done[at_floor] = True
            retry = rows[floor & ~fresh[rows]]
            B[retry] = eye
            fresh[retry] = True
            flat[retry] = 0
    return x, f, dict(converged=converged, stalled=stalled, failed=failed, iterations=it, calls=calls, kink_calls=kink_calls)


def bestfit_batched(lf, points=None, guess=None, livetime_days=None, gtol=1e-6, max_iter=200, return_info=False,
                    multi_start=True, keep_starts=2, scout_iterations=6, also_from=(), datasets=None, **fixed):
    """Maximise `lf` over its floating parameters for P hypotheses at once.

    points: dict parameter name -> array [P] of values held fixed per problem (the scan grid / the hypotheses);
    fixed (kwargs): parameters held at one value in every problem; everything else floats, as in `bestfit_scipy`
    (rate multipliers first, guess 1; then shape parameters, guess = base value; blueice/inference.py:79-102).
    guess: dict name -> scalar or array [P] (e.g. the neighbouring hypothesis' solution).
    multi_start: also start from the other grid cells of every floating shape parameter (see below; 'cells' = from the
    centre of every cell of their anchor grid, the thorough and expensive variant); the reference's single start is
    multi_start=False.  also_from: more starting points, each a dict name -> scalar or array [P] (floating
    parameters it does not name start at their guess) -- e.g. the global best fit's nuisances for a profile fit.
    datasets: array [P] of dataset indices, one per problem (the likelihood holds several datasets: `set_binned_data` with
    a stack, `simulate_toys`) -- every toy of a toy-MC ensemble fitted at the same time instead of the reference's loop
    over `d = simulate(); lf.set_data(d); bestfit_scipy(lf)`; with `points`, problem p is (hypothesis p, dataset p).
    -> (OrderedDict name -> fitted values [P], max log likelihood [P]) [, info]."""
    points = {} if points is None else {k: np.atleast_1d(np.asarray(v, dtype=float)) for k, v in points.items()}
    if datasets is not None:
        datasets = np.atleast_1d(np.asarray(datasets, dtype=np.int64))
    P = max([len(v) for v in points.values()] + [1 if datasets is None else len(datasets)])
    if datasets is not None:
        datasets = np.ascontiguousarray(np.broadcast_to(datasets, (P,)))
    points = {k: np.broadcast_to(v, (P,)) for k, v in points.items()}
    guess = guess or {}
    names, x0, lo, hi = [], [], [], []
    for src in lf.rate_parameters:
        key = '%s_rate_multiplier' % src
        if key in fixed or key in points:
            continue
        names.append(key)
        x0.append(np.broadcast_to(np.asarray(guess.get(key, 1.0), dtype=float), (P,)))
        b = lf.get_bounds(key)
        lo.append(b[0]); hi.append(b[1])
    for key, (_, _, base_value) in lf.shape_parameters.items():
        if key in fixed or key in points:
            continue
        gval = guess.get(key)
        if gval is None:
            gval = lf.pdf_base_config.get(key)
            if not is_numeric(gval):
                gval = base_value
        names.append(key)
        x0.append(np.broadcast_to(np.asarray(gval, dtype=float), (P,)))
        b = lf.get_bounds(key)
        lo.append(b[0]); hi.append(b[1])
    if not names:
        raise NoOpimizationNecessary("There are no parameters to fit, no optimization is necessary")
    x0 = np.stack(x0, axis=1)
    lo, hi = np.array(lo, dtype=float), np.array(hi, dtype=float)
    obj = BatchObjective(lf, names, points, fixed, livetime_days, datasets)
    # interior anchors of the floating shape parameters: where the morph, and with it the likelihood, has a kink
    kinks = []
    for key in names:
        anchors = lf.shape_parameters[key][0] if key in lf.shape_parameters else None
        zs = np.sort(np.array([float(a) for a in (anchors or ()) if is_numeric(a)]))
        kinks.append(zs[1:-1] if len(zs) > 2 else np.zeros(0))
    obj.kinks = kinks
    # Other starting points: the morph is smooth only inside a grid cell of the anchors, and with noisy templates the
    # likelihood can have one local maximum per cell along a shape parameter.  Besides the reference's starting point
    # (the base value) every problem is therefore also started from the centre of each OTHER cell along each floating
    # shape parameter (one axis at a time, not their product); all starts take `scout_iterations` steps together, then
    # the best `keep_starts` per problem run to convergence.  The price is a few more rows per device call.
    starts = [x0]
    thorough = isinstance(multi_start, str) and multi_start == 'cells'
    if thorough:
        # the thorough variant: one start in the centre of EVERY cell of the floating shape parameters' anchor grid (their
        # product: 4^d cells for five anchors per axis) -- for likelihoods with a hump per cell, at that many times the rows
        import itertools
        centres = []
        for j, key in enumerate(names):
            anchors = lf.shape_parameters.get(key, (None,))[0] if key in lf.shape_parameters else None
            zs = np.sort(np.array([float(a) for a in (anchors or ()) if is_numeric(a)]))
            if len(zs) > 1:
                centres.append((j, 0.5 * (zs[:-1] + zs[1:])))
        for combo in itertools.product(*[c for _, c in centres]):
            alt = x0.copy()
            for (j, _), v in zip(centres, combo):
                alt[:, j] = v
            starts.append(alt)
    if multi_start:
        for j, key in enumerate(names):
            anchors = lf.shape_parameters.get(key, (None,))[0] if key in lf.shape_parameters else None
            if not anchors:
                continue
            zs = np.sort(np.array([float(a) for a in anchors if is_numeric(a)]))
            for a, b in zip(zs[:-1], zs[1:]):
                inside = (x0[:, j] > a) & (x0[:, j] < b)         # (a start ON an anchor -- the usual case: base values are
                #  anchors -- is inside neither neighbour: both get a start.  40 % more rows on C2, where it changes nothing;
                #  on the 40-bin test model a quarter of a profiled scan's points end up to 0.5 higher for it)
                if np.all(inside):
                    continue
                alt = x0.copy()
                alt[:, j] = np.where(inside, x0[:, j], 0.5 * (a + b))
                starts.append(alt)
    # A start that sits ON a kink (the base value of a shape parameter is an anchor) from which the function falls to BOTH
    # sides: the kink rule of the optimiser goes down the steeper one, and the other grid cell -- whose maximum may be the
    # higher one -- would never be seen.  Those problems, and only those, get one more start: a hair into the other side
    # (the rows of all other problems in that start are nan and drop out at its first evaluation).
    n_vee = 0
    if multi_start:
        on = np.zeros(x0.shape, dtype=bool)
        for j, ks in enumerate(kinks):
            if len(ks):
                on[:, j] = np.isin(x0[:, j], ks)
        pi, ji = np.nonzero(on)
        if len(pi):
            k = np.arange(len(pi))
            xt = np.concatenate([x0[pi], x0[pi]])
            xt[k, ji] = np.nextafter(xt[k, ji], -np.inf)
            xt[len(pi) + k, ji] = np.nextafter(xt[len(pi) + k, ji], np.inf)
            ft, gt = obj(xt, np.concatenate([pi, pi]))
            gl, gr = gt[k, ji], gt[len(pi) + k, ji]
            with np.errstate(invalid='ignore'):
                vee = np.isfinite(ft[:len(pi)]) & np.isfinite(ft[len(pi):]) & (gl > 0) & (gr < 0)
            for j in np.unique(ji[vee]):
                sel = vee & (ji == j)
                rows = pi[sel]
                other = np.where(gl[sel] > -gr[sel], 1.0, -1.0)        # the optimiser takes the steeper side: start on the other
                alt = np.full_like(x0, np.nan)
                alt[rows] = x0[rows]
                ks = kinks[j]
                at = np.searchsorted(ks, x0[rows, j])
                width = np.where(other > 0, np.append(ks, hi[j])[at + 1] - ks[at], ks[at] - np.append(lo[j], ks)[at])
                alt[rows, j] = x0[rows, j] + other * 1e-6 * np.where(np.isfinite(width), width, 1.0)
                starts.append(alt)
                n_vee += len(rows)
    for extra in also_from:
        alt = x0.copy()
        for j, key in enumerate(names):
            if key in extra:
                alt[:, j] = np.broadcast_to(np.asarray(extra[key], dtype=float), (P,))
        starts.append(alt)
    n_st = len(starts)
    if n_st == 1:
        x, f, info = batched_minimize(obj, x0, lo, hi, gtol=gtol, max_iter=max_iter, kinks=kinks)
    else:
        rows_of = lambda k: np.tile(np.arange(P), k)
        scout = obj.stacked(P)                                       # row r of the stacked problem set is problem r % P
        xs, fs, _ = batched_minimize(scout, np.concatenate(starts), lo, hi, gtol=gtol, max_iter=scout_iterations, kinks=kinks)
        fs = np.where(np.isfinite(fs), fs, np.inf).reshape(n_st, P)
        keep = n_st if thorough else min(keep_starts + len(also_from), n_st)      # ('cells': every start runs to convergence)
        order = np.argsort(fs, axis=0, kind='stable')[:keep]          # [keep, P] start indices, best first
        xk = xs.reshape(n_st, P, -1)[order, np.arange(P)[None, :]].reshape(keep * P, -1)
        x, f, info = batched_minimize(scout, xk, lo, hi, gtol=gtol, max_iter=max_iter, kinks=kinks)
        f2 = np.where(np.isfinite(f), f, np.inf).reshape(keep, P)
        win = np.argmin(f2, axis=0)
        pick = win * P + np.arange(P)
        x, f = x[pick], f[pick]
        info = dict(info, **{k: info[k][pick] for k in ('converged', 'stalled', 'failed')})
        info['starts'] = n_st
        info['vee_starts'] = n_vee
        info['winning_start'] = order[win, np.arange(P)]
        del rows_of
    info['evaluations'] = obj.evaluations
    info['calls'] = obj.calls
    info['analytic_gradient'] = obj.analytic
    best = OrderedDict((n, x[:, j].copy()) for j, n in enumerate(names))
    return (best, -f, info) if return_info else (best, -f)
