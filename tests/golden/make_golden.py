#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL reference.

Development-container only (the reference does not exist on the GPU box).  Run as

    mkdir -p /tmp/golden_scratch && cd /tmp/golden_scratch && PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/repo/tools/oracle_shims:/root/reference:/root/repo/tests \
        python /root/repo/tests/golden/make_golden.py [case names ...]

The likelihoods are built by tests/model_zoo.py from the reference's own classes
(`namespace_of('blueice')`); the same builders later run on blueice_amd in the drop-in tests.
The two third-party modules the reference needs but the image lacks (multihist, atomicwrites) are
replaced by the build-authored stand-ins in tools/oracle_shims (see their docstrings); neither is
on the per-call arithmetic path (a3-a6), and every fixture records the binned tensors so the pinned
quantity is "logL given tensors".

Each fixture `<case>.npz` holds
    d, S, bins                      shape info
    anchor_z_<i>                    sorted anchor z values of shape axis i
    ps      [A.., S, *bins]         anchor PMF tensor   (values of lf.ps_interpolator)
    mus     [A.., S]                anchor expected events (values of lf.mus_interpolator)
    n_model [A.., S, *bins]         anchor MC counts (BB cases only)
    counts  [*bins]                 binned data
    bb_source                       -1 or the Beeston-Barlow source index
    livetime_base                   pdf_base_config['livetime_days'] or nan
    call_z [N, d], call_mult [N, S], call_livetime [N] (nan = not given), call_ll [N]
    call_scale [N, S]               rate multiplier x livetime scaling x efficiency: what multiplies mus
    allow_negative [S]              the sources' allow_negative flags
    call_asserts_<j>                present when the reference raised AssertionError on call j
    full_<j>_mus / full_<j>_ps      `full_output=True` results for call j (a few calls)
fit_c1_like.npz additionally holds bestfit_scipy results (names, values, max logL).
api_<case>.npz (model_zoo.API_CASES: compute_pdf, priors, zero live time) hold only call_ll [N] and call_error [N]
(exception class name or '') -- those calls are not functions of the anchor tensors alone.
"""
import os

import numpy as np

import blueice
import model_zoo

OUT = os.path.dirname(os.path.abspath(__file__))


def tensors_of_unbinned(lf):
    """Unbinned: ps = pdf of every source at every event for every anchor [A.., S, N]; no counts."""
    d = len(lf.shape_parameters)
    out = {}
    if d:
        rgi_ps = lf.ps_interpolator.__closure__[0].cell_contents
        rgi_mu = lf.mus_interpolator.__closure__[0].cell_contents
        for i, g in enumerate(rgi_ps.grid):
            out['anchor_z_%d' % i] = np.asarray(g, dtype=float)
        out['ps'] = np.asarray(rgi_ps.values, dtype=float)
        out['mus'] = np.asarray(rgi_mu.values, dtype=float)
    else:
        out['ps'] = np.asarray(lf.ps, dtype=float)
        out['mus'] = np.asarray(lf.base_model.expected_events(), dtype=float)
    out['d'] = d
    out['S'] = len(lf.source_name_list)
    out['bins'] = np.array([out['ps'].shape[-1]])
    out['counts'] = np.zeros(0)
    out['bb_source'] = -1
    out['kind'] = 1
    out['outlier'] = float(lf.config.get('outlier_likelihood', 1e-12))
    out['livetime_base'] = float(lf.pdf_base_config.get('livetime_days', np.nan))
    return out


def tensors_of(lf):
    """Pull the anchor tensors out of a prepared reference likelihood."""
    if type(lf).__name__ == 'UnbinnedLogLikelihood':
        return tensors_of_unbinned(lf)
    d = len(lf.shape_parameters)
    out = {}
    if d:
        rgi_ps = lf.ps_interpolator.__closure__[0].cell_contents
        rgi_mu = lf.mus_interpolator.__closure__[0].cell_contents
        for i, g in enumerate(rgi_ps.grid):
            out['anchor_z_%d' % i] = np.asarray(g, dtype=float)
        out['ps'] = np.asarray(rgi_ps.values, dtype=float)
        out['mus'] = np.asarray(rgi_mu.values, dtype=float)
        if lf.model_statistical_uncertainty_handling is not None:
            out['n_model'] = np.asarray(
                lf.n_model_events_interpolator.__closure__[0].cell_contents.values, dtype=float)
    else:
        out['ps'] = np.asarray(lf.ps, dtype=float)
        out['mus'] = np.asarray(lf.base_model.expected_events(), dtype=float)
        if lf.model_statistical_uncertainty_handling is not None:
            out['n_model'] = np.asarray(lf.n_model_events, dtype=float)
    out['d'] = d
    out['S'] = len(lf.source_name_list)
    out['bins'] = np.array(lf.ps.shape[1:])
    out['counts'] = np.asarray(lf.data_events_per_bin.histogram, dtype=float)
    bb = -1
    if lf.model_statistical_uncertainty_handling == 'bb_single':
        bb = lf.base_model.get_source_i(lf.config['bb_single_source'])
    out['bb_source'] = bb
    out['livetime_base'] = float(lf.pdf_base_config.get('livetime_days', np.nan))
    return out


def dump(name, lf, calls, full=()):
    t = tensors_of(lf)
    shape_names = list(lf.shape_parameters.keys())
    zs, mults, lts, lls, scales = [], [], [], [], []
    for j, kw in enumerate(calls):
        kw = dict(kw)
        lt = kw.get('livetime_days', np.nan)
        try:
            ll = lf(**kw)
        except AssertionError:
            ll = np.nan
            t['call_asserts_%d' % j] = 1
        mult, settings = lf._kwargs_to_settings(**{k: v for k, v in kw.items() if k != 'livetime_days'})
        zs.append([settings[n] for n in shape_names])
        mults.append(mult)
        sc = np.array(mult, dtype=float)
        if not (isinstance(lt, float) and np.isnan(lt)):
            sc = sc * (lt / lf.pdf_base_config['livetime_days'])
        for i_s, (use, en) in enumerate(zip(lf.source_apply_efficiency, lf.source_efficiency_names)):
            if use:
                sc[i_s] *= settings.get(en, 1)
        scales.append(sc)
        lts.append(lt)
        lls.append(ll)
        if j in full and np.isfinite(ll):
            r, m, p = lf(full_output=True, **kw)
            t['full_%d_mus' % j] = np.asarray(m, dtype=float)
            t['full_%d_ps' % j] = np.asarray(p, dtype=float)
    t['call_z'] = np.asarray(zs, dtype=float).reshape(len(calls), len(shape_names))
    t['call_mult'] = np.asarray(mults, dtype=float)
    t['call_livetime'] = np.asarray(lts, dtype=float)
    t['call_ll'] = np.asarray(lls, dtype=float)
    t['call_scale'] = np.asarray(scales, dtype=float)      # multiplier x livetime x efficiency (likelihood.py:366-393)
    t['allow_negative'] = np.array([1 if a else 0 for a in lf.source_allowed_negative])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **t)
    print('%-28s d=%d S=%d bins=%s calls=%d  ll[0]=%r' % (
        name, t['d'], t['S'], [int(b) for b in t['bins']], len(calls), float(lls[0])))


def fit_goldens(ns):
    """(inputs -> best-fit dict, max logL) tuples of bestfit_scipy for the caller test."""
    lf = model_zoo.fit_c1_like(ns)
    t = tensors_of(lf)
    res, ll = lf.bestfit_scipy()
    t['fit_all_names'] = np.array(list(res.keys()))
    t['fit_all_values'] = np.array(list(res.values()), dtype=float)
    t['fit_all_ll'] = ll
    res, ll = lf.bestfit_scipy(shift=0.2)
    t['fit_fixshift_names'] = np.array(list(res.keys()))
    t['fit_fixshift_values'] = np.array(list(res.values()), dtype=float)
    t['fit_fixshift_ll'] = ll
    res, ll = lf.bestfit_scipy(shift=0.2, s0_rate_multiplier=1., s1_rate_multiplier=0.9)
    t['fit_none_ll'] = ll
    t['upper_s1_90'] = lf.one_parameter_interval('s1_rate_multiplier', bound=50., kind='upper', confidence_level=0.9)
    t['central_s0_68'] = np.array(lf.one_parameter_interval('s0_rate_multiplier', bound=(2., 20.), kind='central',
                                                             confidence_level=0.68, s1_rate_multiplier=0.))
    np.savez_compressed(os.path.join(OUT, 'fit_c1_like.npz'), **t)
    print('fit_c1_like', dict(zip(t['fit_all_names'], t['fit_all_values'])), t['fit_all_ll'])


def dump_api(name, lf, calls):
    """Return values / exception names of calls that are not functions of the anchor tensors alone."""
    lls, errs = [], []
    for kw in calls:
        try:
            lls.append(float(lf(**kw)))
            errs.append('')
        except (AssertionError, ValueError, NotImplementedError) as e:
            lls.append(np.nan)
            errs.append(type(e).__name__)
    extra = {}
    if getattr(lf, 'source_wise_interpolation', False) and hasattr(lf, 'ps_interpolators'):
        # source-wise interpolation: every source's OWN anchor tensor (pdf at the events, expected events), as the
        # reference's per-source interpolators hold them -- the host layer must upload their full-grid expansion
        for i, sn in enumerate(lf.source_name_list):
            it = lf.ps_interpolators[sn]
            if callable(it):
                rgi = it.__closure__[0].cell_contents
                extra['sw_%d_ps' % i] = np.asarray(rgi.values, dtype=float)
                for k, g in enumerate(rgi.grid):
                    extra['sw_%d_z_%d' % (i, k)] = np.asarray(g, dtype=float)
                extra['sw_%d_dims' % i] = np.array(lf._get_shape_indices(sn))
                extra['sw_%d_mus' % i] = np.array([src.expected_events for src in lf.anchor_sources[sn].values()],
                                                  dtype=float).reshape(rgi.values.shape[:-1])
            else:
                extra['sw_%d_ps' % i] = np.asarray(it, dtype=float)
                extra['sw_%d_dims' % i] = np.array([], dtype=int)
                extra['sw_%d_mus' % i] = np.array(lf.base_model.sources[i].expected_events, dtype=float)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), call_ll=np.array(lls), call_error=np.array(errs), **extra)
    print('%-28s calls=%d  %s' % (name, len(calls), [e or round(v, 6) for v, e in zip(lls, errs)]))


if __name__ == '__main__':
    import scipy
    print('reference blueice', blueice.__version__, 'numpy', np.__version__, 'scipy', scipy.__version__)
    import sys
    only = set(sys.argv[1:])              # optional: names of the cases to (re)generate; default all
    ns = model_zoo.namespace_of('blueice')
    for name, builder in list(model_zoo.CASES.items()) + list(model_zoo.UNBINNED_CASES.items()):
        if only and name not in only:
            continue
        lf, calls, full = builder(ns)
        dump(name, lf, calls, full)
    for name, builder in model_zoo.API_CASES.items():
        if only and name not in only:
            continue
        lf, calls = builder(ns)
        dump_api(name, lf, calls)
    if not only or 'fit' in only:
        fit_goldens(ns)
