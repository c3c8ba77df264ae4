#!/usr/bin/env python3
"""Golden vectors for PROFILED likelihood scans and intervals, from the REAL reference (development container only):

    mkdir -p /tmp/golden_scratch && cd /tmp/golden_scratch && PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/repo/tools/oracle_shims:/root/reference:/root/repo/tests \
        python /root/repo/tests/golden/make_golden_profile.py [scan names ...]

For every scan of model_zoo.PROFILE_SCANS the reference's own `bestfit_scipy` (blueice/inference.py:131-178) is run at
every grid point with the scanned parameters fixed and everything else floating -- the double loop of
`plot_likelihood_ratio` (:424-432) -- and the per-point maxima and best-fit values are stored in
tests/golden/profile_<name>.npz:
    axis_<i>_name / axis_<i>_values     the scan space
    fixed_names / fixed_values           parameters held fixed through kwargs
    float_names                          the fitted parameters, in the reference's order
    ll [grid]                            max log likelihood per grid point (bestfit_scipy with minimize_kwargs = TIGHT below; nan where
                                         the reference then raises OptimizationFailed)
    ll_default [grid]                    the same with the reference's default minimiser settings
    best [grid, F]                       fitted values per grid point
    global_names / global_values / global_ll     the unconstrained best fit (denominator of the ratio)
plus, for the first model, intervals of `one_parameter_interval` (:332-389).  The drop-in's batched profile-fit engine
(blueice_amd/profile.py) is compared with these numbers in tests/test_profile_gpu.py.
"""
import os
import sys
import time

import numpy as np

import blueice
import model_zoo

OUT = os.path.dirname(os.path.abspath(__file__))
TIGHT = {'tol': 1e-10, 'options': {'maxiter': 40000}}


def main():
    only = set(sys.argv[1:])
    ns = model_zoo.namespace_of('blueice')
    built = {}
    for name, (builder, space, fixed) in model_zoo.PROFILE_SCANS.items():
        if only and name not in only:
            continue
        lf = built.get(builder)
        if lf is None:
            lf = built[builder] = builder(ns)
        t0 = time.time()
        names = [n for n, _ in space]
        grids = np.meshgrid(*[np.asarray(v, dtype=float) for _, v in space], indexing='ij')
        ll = np.empty(grids[0].shape)
        best = None
        float_names = None
        # twice: with the reference's default minimiser settings (what a user gets: scipy's BFGS on numeric differences
        # stops ~1e-5 short of the maximum), and with tol = 1e-10 -- still the reference's own bestfit_scipy; its BFGS then
        # ends in "precision loss" and the reference's Nelder-Mead fallback (inference.py:157-167) finishes the job
        # to ~1e-9 -- which is what a fit that really reaches the maximum has to agree with
        ll_default = np.empty(grids[0].shape)
        for idx in np.ndindex(*grids[0].shape):
            kw = dict(fixed, **{n: float(g[idx]) for n, g in zip(names, grids)})
            ll_default[idx] = lf.bestfit_scipy(**kw)[1]
            try:
                res, val = lf.bestfit_scipy(minimize_kwargs=TIGHT, **kw)
            except blueice.exceptions.OptimizationFailed:          # (a maximum on the edge of the allowed region, where the
                res, val = lf.bestfit_scipy(**kw)                   # objective is +inf next door: Nelder-Mead gives up)
                val = np.nan
            if best is None:
                float_names = list(res.keys())
                best = np.empty(grids[0].shape + (len(float_names),))
            ll[idx] = val
            best[idx] = [res[k] for k in float_names]
        gres, gll = lf.bestfit_scipy(minimize_kwargs=TIGHT, **fixed)
        t = dict(ll=ll, ll_default=ll_default, best=best, float_names=np.array(float_names), fixed_names=np.array(list(fixed.keys())),
                 fixed_values=np.array(list(fixed.values()), dtype=float), global_names=np.array(list(gres.keys())),
                 global_values=np.array(list(gres.values()), dtype=float), global_ll=gll,
                 global_ll_default=lf.bestfit_scipy(**fixed)[1])
        for i, (n, v) in enumerate(space):
            t['axis_%d_name' % i] = np.array(n)
            t['axis_%d_values' % i] = np.asarray(v, dtype=float)
        if name == 'd2_rate_160':
            tight = dict(minimize_kwargs=TIGHT)
            fit = lambda lf_, **kw: lf_.bestfit_scipy(**dict(tight, **kw))
            t['upper_s0_90'] = lf.one_parameter_interval('s0_rate_multiplier', bound=40., kind='upper', confidence_level=0.9,
                                                         bestfit_routine=fit, **fixed)
            t['upper_s1_95'] = lf.one_parameter_interval('s1_rate_multiplier', bound=20., kind='upper', confidence_level=0.95,
                                                         bestfit_routine=fit, **fixed)
        np.savez_compressed(os.path.join(OUT, 'profile_%s.npz' % name), **t)
        print('%-24s %d fits in %.0f s; ll in [%.6f, %.6f], global %.9f; default settings differ by up to %.2e' % (
            name, ll.size, time.time() - t0, np.nanmin(ll), np.nanmax(ll), gll, np.nanmax(np.abs(ll - ll_default))), flush=True)


if __name__ == '__main__':
    import scipy
    print('reference blueice', blueice.__version__, 'numpy', np.__version__, 'scipy', scipy.__version__, flush=True)
    main()
