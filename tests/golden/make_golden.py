#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL reference.

Development-container only (the reference does not exist on the GPU box).  Run as

    cd /tmp/golden_scratch && PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/repo/tools/oracle_shims:/root/reference \
        python /root/repo/tests/golden/make_golden.py

The two third-party modules the reference needs but the image lacks (multihist, atomicwrites)
are replaced by the build-authored stand-ins in tools/oracle_shims (see their docstrings);
neither is on the per-call arithmetic path (a3-a6), and every fixture records the binned
tensors so the pinned quantity is "logL given tensors".

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
    full_<j>_mus / full_<j>_ps      `full_output=True` results for call j (a few calls)
"""
import os
import sys
from collections import OrderedDict

import numpy as np

import blueice
from blueice.likelihood import BinnedLogLikelihood
from blueice.source import DensityEstimatingSource
from blueice.test_helpers import (FixedSampleSource, GaussianMCSource, conf_for_test, make_data)

OUT = os.path.dirname(os.path.abspath(__file__))


class MorphedSampleSource(DensityEstimatingSource):
    """FixedSampleSource whose sample and rate respond to numeric shape settings:
    events are shifted/scaled by (shift, stretch, tilt) with source-specific strengths."""

    def __init__(self, config, *args, **kwargs):
        super().__init__(config, *args, **kwargs)
        k = self.config.get('strength', 1.0)
        self.events_per_day *= (1 + 0.05 * k * self.config.get('shift', 0.)
                                - 0.03 * k * self.config.get('stretch', 0.)
                                + 0.02 * k * self.config.get('tilt', 0.))

    def get_events_for_density_estimate(self):
        d = self.config['data'].copy()
        k = self.config.get('strength', 1.0)
        names = [n for n, _ in self.config['analysis_space']]
        d[names[0]] = d[names[0]] + 0.31 * k * self.config.get('shift', 0.)
        if len(names) > 1:
            d[names[1]] = d[names[1]] * (1 + 0.11 * k * self.config.get('stretch', 0.))
        if len(names) > 2:
            d[names[2]] = d[names[2]] + 0.07 * k * self.config.get('tilt', 0.) * d[names[0]]
        return d, len(d)


def tensors_of(lf):
    """Pull the anchor tensors out of a prepared reference likelihood."""
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
    """calls: list of kwargs dicts for lf(**kw).  full: indices of calls to also dump full_output."""
    t = tensors_of(lf)
    shape_names = list(lf.shape_parameters.keys())
    zs, mults, lts, lls = [], [], [], []
    for j, kw in enumerate(calls):
        kw = dict(kw)
        lt = kw.get('livetime_days', np.nan)
        try:
            ll = lf(**kw)
        except AssertionError:
            ll = np.nan       # recorded as "reference asserts here"
            t['call_asserts_%d' % j] = 1
        mult, settings = lf._kwargs_to_settings(**{k: v for k, v in kw.items() if k != 'livetime_days'})
        zs.append([settings[n] for n in shape_names])
        mults.append(mult)
        lts.append(lt if lt is not None else np.nan)
        lls.append(ll)
        if j in full and np.isfinite(ll):
            r, m, p = lf(full_output=True, **kw)
            t['full_%d_mus' % j] = np.asarray(m, dtype=float)
            t['full_%d_ps' % j] = np.asarray(p, dtype=float)
    t['call_z'] = np.asarray(zs, dtype=float).reshape(len(calls), len(shape_names))
    t['call_mult'] = np.asarray(mults, dtype=float)
    t['call_livetime'] = np.asarray(lts, dtype=float)
    t['call_ll'] = np.asarray(lls, dtype=float)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **t)
    print('%-28s d=%d S=%d bins=%s calls=%d  ll[0]=%r' % (
        name, t['d'], t['S'], list(t['bins']), len(calls), lls[0]))


def sample(rng, n, space):
    """n events uniformly-ish inside the analysis space, clustered so bins are unevenly filled."""
    d = np.zeros(n, dtype=[('source', int)] + [(nm, float) for nm, _ in space])
    for nm, edges in space:
        lo, hi = edges[0], edges[-1]
        u = rng.beta(2.0, 3.5, size=n)
        d[nm] = lo + (hi - lo) * (0.02 + 0.96 * u)
    return d


# ---------------------------------------------------------------------------------------------
# (1) the reference's own tests, restated as fixtures
# ---------------------------------------------------------------------------------------------
def ref_single_bin():
    np.random.seed(1)
    conf = conf_for_test(mc=True, analysis_space=[['x', [-40, 40]]])
    lf = BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(np.zeros(1, dtype=[('x', float), ('source', int)]))
    dump('ref_single_bin', lf, [{}, dict(s0_rate_multiplier=5.4), dict(s0_rate_multiplier=0.)], full=(1,))


def ref_zero_bin():
    np.random.seed(2)
    conf = conf_for_test(mc=True, analysis_space=[['x', [-40, 40]]])
    lf = BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(np.zeros(0, dtype=[('x', float), ('source', int)]))
    dump('ref_zero_bin', lf, [dict(s0_rate_multiplier=0.), {}, dict(s0_rate_multiplier=2.)])


def ref_multi_bin_single_dim():
    data, n_mc = make_data([dict(n_events=24, x=0.5), dict(n_events=56, x=1.5)])
    conf = conf_for_test(events_per_day=42, analysis_space=[['x', [0, 1, 5]]],
                         default_source_class=FixedSampleSource, data=data)
    lf = BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    data, _ = make_data([dict(n_events=18, x=0.5), dict(n_events=70, x=1.5)])
    lf.set_data(data)
    dump('ref_multi_bin_single_dim', lf, [{}, dict(s0_rate_multiplier=1.7)])


def ref_multi_bin():
    data, n_mc = make_data([dict(n_events=24, x=0.5, y=0.5), dict(n_events=56, x=1.5, y=0.5),
                            dict(n_events=6, x=0.5, y=2), dict(n_events=14, x=1.5, y=2)])
    conf = conf_for_test(events_per_day=42, default_source_class=FixedSampleSource, data=data,
                         analysis_space=[['x', [0, 1, 5]], ['y', [0, 1, 4]]])
    lf = BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi', 3: 'wha'}, base_value=1)
    lf.prepare()
    data, _ = make_data([dict(n_events=18, x=0.5, y=0.5), dict(n_events=70, x=1.5, y=0.5),
                         dict(n_events=4, x=0.5, y=2), dict(n_events=10, x=1.5, y=2)])
    lf.set_data(data)
    dump('ref_multi_bin', lf,
         [dict(strlen_multiplier=1), dict(strlen_multiplier=2), dict(strlen_multiplier=2.3),
          dict(strlen_multiplier=3), dict(strlen_multiplier=1.00001, s0_rate_multiplier=0.3),
          dict(strlen_multiplier=3.5), dict(strlen_multiplier=0.99)], full=(2,))


BB_LC = {'model_statistical_uncertainty_handling': 'bb_single', 'bb_single_source': 0}


def ref_bb_single_bin():
    data, n_mc = make_data([dict(n_events=32, x=0.5)])
    conf = conf_for_test(default_source_class=FixedSampleSource, events_per_day=32/5,
                         analysis_space=[['x', [0, 1]]], data=data)
    lf = BinnedLogLikelihood(conf, likelihood_config=dict(BB_LC))
    lf.prepare()
    lf.set_data(np.zeros(2, dtype=[('x', float), ('source', int)]))
    dump('ref_bb_single_bin', lf, [{}], full=(0,))


def ref_bb_multi_bin():
    data, n_mc = make_data([dict(n_events=16, x=0.5), dict(n_events=30, x=1.5),
                            dict(n_events=32, x=2.5), dict(n_events=27, x=3.5)])
    conf = conf_for_test(default_source_class=FixedSampleSource, events_per_day=105/5,
                         analysis_space=[['x', [0, 1, 2, 3, 4]]], data=data)
    lf = BinnedLogLikelihood(conf, likelihood_config=dict(BB_LC))
    lf.prepare()
    data, _ = make_data([dict(n_events=3, x=0.5), dict(n_events=5, x=1.5),
                         dict(n_events=2, x=2.5), dict(n_events=7, x=3.5)])
    lf.set_data(data)
    dump('ref_bb_multi_bin', lf, [{}], full=(0,))


def _bb_second_source_lf(extra=False):
    cal, _ = make_data([dict(n_events=16, x=0.5), dict(n_events=30, x=1.5),
                        dict(n_events=32, x=2.5), dict(n_events=27, x=3.5)])
    oth, _ = make_data([dict(n_events=5, x=0.5), dict(n_events=7, x=1.5),
                        dict(n_events=1, x=2.5), dict(n_events=3, x=3.5)])
    conf = conf_for_test(default_source_class=FixedSampleSource,
                         analysis_space=[['x', [0, 1, 2, 3, 4]]], dummy=1)
    conf['sources'] = [{'name': 's0', 'events_per_day': 105/5., 'data': cal},
                       {'name': 's1', 'events_per_day': 16., 'data': oth}]
    lf = BinnedLogLikelihood(conf, likelihood_config=dict(BB_LC))
    lf.add_shape_parameter('dummy', (0, 1))
    if extra:
        lf.add_rate_parameter('s1')
        lf.add_rate_parameter('s0')
        lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi', 3: 'wha'}, base_value=1)
    lf.prepare()
    data, _ = make_data([dict(n_events=3, x=0.5), dict(n_events=5, x=1.5),
                         dict(n_events=2, x=2.5), dict(n_events=7, x=3.5)])
    lf.set_data(data)
    return lf


def ref_bb_second_source():
    dump('ref_bb_second_source', _bb_second_source_lf(), [{}, dict(dummy=0.25), dict(dummy=0)], full=(0,))


def bb_two_shape():
    lf = _bb_second_source_lf(extra=True)
    dump('bb_two_shape', lf,
         [{}, dict(strlen_multiplier=1.5, dummy=.25), dict(strlen_multiplier=2.3, dummy=.5, s1_rate_multiplier=.8),
          dict(strlen_multiplier=3, dummy=1, s0_rate_multiplier=1.3), dict(strlen_multiplier=2, dummy=0.999),
          dict(s1_rate_multiplier=0.), dict(strlen_multiplier=2.75, dummy=0.1, s0_rate_multiplier=0.4,
                                            s1_rate_multiplier=2.2)], full=(2, 5))


# ---------------------------------------------------------------------------------------------
# (2) FixedSampleSource-style models with numeric shape parameters
# ---------------------------------------------------------------------------------------------
def _morph_lf(rng, S, space, shape_anchors, n_mc, n_data, lc=None, livetime=None, bb_floor=False):
    strengths = [1.0, -0.6, 0.45, 1.7, -1.2, 0.8]
    conf = dict(sources=[], default_source_class=MorphedSampleSource,
                analysis_space=space, force_recalculation=True, never_save_to_cache=True,
                shift=0., stretch=0., tilt=0.)
    if livetime is not None:
        conf['livetime_days'] = livetime
    for s in range(S):
        d = sample(rng, n_mc, space)
        if bb_floor and s == 0:
            # guarantee >= 1 MC event in every bin for every anchor: add a lattice of bin centres
            centres = np.meshgrid(*[0.5 * (np.asarray(e)[1:] + np.asarray(e)[:-1]) for _, e in space], indexing='ij')
            lat = np.zeros(centres[0].size * 3, dtype=d.dtype)
            for (nm, _), c in zip(space, centres):
                lat[nm] = np.tile(c.ravel(), 3)
            d = np.concatenate([d, lat])
        conf['sources'].append(dict(name='s%d' % s, events_per_day=40. * (s + 1), data=d,
                                    strength=0.0 if (bb_floor and s == 0) else strengths[s]))
    lf = BinnedLogLikelihood(conf, likelihood_config=lc)
    for s in range(S):
        lf.add_rate_parameter('s%d' % s)
    for nm, anchors in shape_anchors.items():
        lf.add_shape_parameter(nm, anchors)
    lf.prepare()
    lf.set_data(sample(rng, n_data, space))
    return lf


def c1_like():
    rng = np.random.default_rng(11)
    space = [['x', np.linspace(-4, 4, 41)]]
    lf = _morph_lf(rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 4000, 300)
    calls = [{}, dict(shift=-1.), dict(shift=1.), dict(shift=0.37), dict(shift=-0.82, s0_rate_multiplier=1.4),
             dict(shift=0.999999, s1_rate_multiplier=0.), dict(shift=1.2), dict(shift=-1.0000001),
             dict(s0_rate_multiplier=-0.1), dict(shift=0.5, s0_rate_multiplier=0., s1_rate_multiplier=0.)]
    dump('c1_like', lf, calls, full=(3,))


def d2_nonuniform():
    rng = np.random.default_rng(12)
    space = [['x', np.array([-3., -1.5, -0.5, 0., 0.4, 1.1, 3.])], ['y', np.linspace(0, 5, 6)]]
    lf = _morph_lf(rng, 3, space, OrderedDict(shift=(-1., -0.25, 0.5, 2.), stretch=(0., 1., 4.)),
                   3000, 500, livetime=2.)
    calls = [{}]
    for z0 in (-1., -0.6, -0.25, 0.1, 0.5, 1.3, 2.):
        for z1 in (0., 0.5, 1., 2.5, 4.):
            calls.append(dict(shift=z0, stretch=z1))
    calls += [dict(shift=0.3, stretch=3.3, s0_rate_multiplier=0.5, s1_rate_multiplier=2., s2_rate_multiplier=1.1),
              dict(shift=0.3, stretch=3.3, livetime_days=5.), dict(shift=2.01, stretch=1.),
              dict(shift=0., stretch=-0.01), dict(shift=1., stretch=2., s2_rate_multiplier=-1.)]
    dump('d2_nonuniform', lf, calls, full=(9, 36))


def d3_small():
    rng = np.random.default_rng(13)
    space = [['x', np.linspace(-3, 3, 7)], ['y', np.linspace(0, 5, 6)], ['w', np.linspace(-2, 2, 5)]]
    lf = _morph_lf(rng, 4, space, OrderedDict(shift=(-1., 0., 1.), stretch=(-1., 0., 1.), tilt=(-1., 0., 1.)),
                   5000, 800)
    calls = [{}]
    pts = rng.uniform(-1, 1, size=(16, 3))
    special = [(-1, -1, -1), (1, 1, 1), (0, 0, 0), (1, -1, 0), (0.5, 1, -1), (-1, 0.25, 1), (1, 0, 0.75),
               (0, 0, 1e-9), (-1e-12, 0, 0)]
    for p in list(pts) + [np.array(s, dtype=float) for s in special]:
        calls.append(dict(shift=float(p[0]), stretch=float(p[1]), tilt=float(p[2])))
    calls += [dict(shift=.2, stretch=-.7, tilt=.4, s0_rate_multiplier=1.2, s1_rate_multiplier=0.,
                   s2_rate_multiplier=0.7, s3_rate_multiplier=3.),
              dict(shift=1.0000001, stretch=0, tilt=0), dict(shift=float('nan'), stretch=0, tilt=0)]
    dump('d3_small', lf, calls, full=(5, 26))


def d0_multi_source():
    rng = np.random.default_rng(14)
    space = [['x', np.linspace(-3, 3, 13)], ['y', np.linspace(0, 5, 4)]]
    lf = _morph_lf(rng, 3, space, OrderedDict(), 2000, 150)
    dump('d0_multi_source', lf, [{}, dict(s0_rate_multiplier=2.), dict(s1_rate_multiplier=0., s2_rate_multiplier=.5)],
         full=(1,))


def edge_mu_zero():
    """Bins where every source has p = 0 but data is present -> -inf; empty data; all-zero rates."""
    data, _ = make_data([dict(n_events=10, x=0.5), dict(n_events=30, x=2.5)])
    conf = conf_for_test(events_per_day=20, analysis_space=[['x', [0, 1, 2, 3]]],
                         default_source_class=FixedSampleSource, data=data)
    lf = BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi'}, base_value=1)
    lf.prepare()
    d_hit, _ = make_data([dict(n_events=3, x=0.5), dict(n_events=1, x=1.5)])
    lf.set_data(d_hit)
    dump('edge_mu_zero_hit', lf, [{}, dict(strlen_multiplier=1.5), dict(s0_rate_multiplier=0.)])
    d_ok, _ = make_data([dict(n_events=3, x=0.5), dict(n_events=11, x=2.5)])
    lf.set_data(d_ok)
    dump('edge_mu_zero_ok', lf, [{}, dict(strlen_multiplier=1.5), dict(s0_rate_multiplier=0.)])
    lf.set_data(make_data([])[0])
    dump('edge_empty_data', lf, [{}, dict(strlen_multiplier=2), dict(s0_rate_multiplier=0.)])


def bb_d2():
    rng = np.random.default_rng(15)
    space = [['x', np.linspace(-3, 3, 7)], ['y', np.linspace(0, 5, 4)]]
    lf = _morph_lf(rng, 3, space, OrderedDict(shift=(-1., 0., 1.), stretch=(0., 1.)), 2500, 400,
                   lc=dict(BB_LC), bb_floor=True)
    calls = [{}]
    for p in rng.uniform(0, 1, size=(10, 2)):
        calls.append(dict(shift=float(2 * p[0] - 1), stretch=float(p[1])))
    calls += [dict(shift=-1., stretch=1.), dict(shift=.4, stretch=.6, s0_rate_multiplier=1.5, s1_rate_multiplier=.2),
              dict(shift=.4, stretch=.6, s1_rate_multiplier=0., s2_rate_multiplier=0.)]
    dump('bb_d2', lf, calls, full=(4, 12))


def fit_goldens():
    """(inputs -> best-fit dict, max logL) tuples of bestfit_scipy for the caller test."""
    rng = np.random.default_rng(21)
    space = [['x', np.linspace(-4, 4, 41)]]
    lf = _morph_lf(rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 4000, 300)
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
    # the raw material the test needs to rebuild the same model through the product's host layer
    t['space_x'] = space[0][1]
    for s, src in enumerate(lf.base_model.sources):
        t['mc_x_%d' % s] = src.config['data']['x']
    t['data_x'] = lf._data['x']
    np.savez_compressed(os.path.join(OUT, 'fit_c1_like.npz'), **t)
    print('fit_c1_like', dict(zip(t['fit_all_names'], t['fit_all_values'])), t['fit_all_ll'])


if __name__ == '__main__':
    print('reference blueice', blueice.__version__, 'numpy', np.__version__)
    import scipy
    print('scipy', scipy.__version__)
    for f in (ref_single_bin, ref_zero_bin, ref_multi_bin_single_dim, ref_multi_bin, ref_bb_single_bin,
              ref_bb_multi_bin, ref_bb_second_source, bb_two_shape, c1_like, d2_nonuniform, d3_small,
              d0_multi_source, edge_mu_zero, bb_d2, fit_goldens):
        f()
