#!/usr/bin/env python3
"""Full-size golden scalars from the REAL reference.

Development-container only (the reference does not exist on the GPU box).  Run as

    mkdir -p /tmp/golden_scratch && cd /tmp/golden_scratch && PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/repo/tools/oracle_shims:/root/reference:/root/repo \
        python /root/repo/tests/golden/make_golden_fullsize.py [C2] [C5-2anchor]

The reference's own `BinnedLogLikelihood` (blueice/likelihood.py:576-675) is prepared at BASELINE.json's sizes
through its ordinary plug-in route: a `blueice.source.Source` subclass whose `get_pmf_grid()`
(blueice/source.py:266-267 is the method it stands in for) returns the deterministic synthetic templates of
`blueice_amd.synthetic.SyntheticModel` -- tensors only, regenerated on the GPU box from the same seeds.  Data are
set through the reference's `set_data` (events at bin centres, one per count).  Then `lf(**kw)` of the reference
is called at a handful of parameter points and the 17-digit results are written to
tests/golden/fullsize.json together with the seeds and a few input checksums (so that a mismatch caused by a
different random stream on the box is told apart from a wrong likelihood).

Nothing of tensor size is committed: only seeds, points and scalars.
"""
import json
import os
import sys
import time

import numpy as np
import scipy

import blueice
from blueice.likelihood import BinnedLogLikelihood
from blueice.source import Source

from blueice_amd.synthetic import CONFIGS, SyntheticModel

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fullsize.json')
_MODELS = {}


def synth_of(name, seed, bb_source):
    key = (name, seed, bb_source)
    if key not in _MODELS:
        _MODELS[key] = SyntheticModel.named(name, seed=seed, bb_source=bb_source)
    return _MODELS[key]


class ReferenceTemplateSource(Source):
    """Plug-in source for the REFERENCE: templates of a SyntheticModel at the anchor named by the shape settings."""

    def __init__(self, config, *args, **kwargs):
        synth = synth_of(config['synth_name'], config['synth_seed'], config['synth_bb'])
        z = [float(config['shape%d' % i]) for i in range(synth.d)]
        index = [int(np.flatnonzero(g == zi)[0]) for g, zi in zip(synth.anchor_z, z)]
        self.synth = synth
        self.anchor = int(np.ravel_multi_index(tuple(index), synth.n_anchor))
        self.index = int(config['source_index'])
        config = dict(config, events_per_day=float(synth.anchor_mus(self.anchor)[self.index]))
        super().__init__(config, *args, **kwargs)

    def get_pmf_grid(self):
        ps = self.synth.anchor_ps_cached(self.anchor)[self.index].reshape(self.synth.bins)
        if self.synth.bb_source >= 0:
            return ps, self.synth.anchor_n_model(self.anchor).reshape(self.synth.bins)
        return ps, np.full(self.synth.bins, np.inf)


def reference_likelihood(name, seed, bb_source):
    synth = synth_of(name, seed, bb_source)
    names = ['shape%d' % i for i in range(synth.d)]
    config = dict(
        analysis_space=[('x%d' % i, np.arange(b + 1, dtype=float)) for i, b in enumerate(synth.bins)],
        default_source_class=ReferenceTemplateSource, synth_name=name, synth_seed=seed, synth_bb=bb_source,
        livetime_days=1, never_save_to_cache=True, force_recalculation=True,
        sources=[dict(name='s%d' % s, source_index=s) for s in range(synth.S)],
        **{n: float(g[len(g) // 2]) for n, g in zip(names, synth.anchor_z)})
    lc = {}
    if bb_source >= 0:
        lc = dict(model_statistical_uncertainty_handling='bb_single', bb_single_source=bb_source)
    lf = BinnedLogLikelihood(config, likelihood_config=lc)
    for s in range(synth.S):
        lf.add_rate_parameter('s%d' % s)
    for n, g in zip(names, synth.anchor_z):
        lf.add_shape_parameter(n, tuple(float(v) for v in g))
    lf.prepare()
    return synth, lf


def events_of(synth, counts):
    """One event at the centre of its bin per count: what the reference's set_data bins back into `counts`."""
    flat = np.flatnonzero(counts)
    rep = np.repeat(flat, counts[flat].astype(np.int64))
    multi = np.unravel_index(rep, synth.bins)
    d = np.zeros(len(rep), dtype=[('x%d' % i, float) for i in range(len(synth.bins))])
    for i, m in enumerate(multi):
        d['x%d' % i] = m + 0.5
    return d


def points_of(synth):
    """(label, z, rate multipliers): off-grid default, bottom corner (on anchors), top edge, a switched-off source,
    a point on an interior anchor plane."""
    z0, r0 = synth.default_point()
    lo = np.array([g[0] for g in synth.anchor_z])
    hi = np.array([g[-1] for g in synth.anchor_z])
    mid = np.array([g[len(g) // 2] for g in synth.anchor_z])
    r_off = r0.copy()
    r_off[1] = 0.0
    z_mixed = z0.copy()
    z_mixed[0] = mid[0]
    pts = [('default_offgrid', z0, r0), ('bottom_corner', lo, np.ones(synth.S)), ('top_edge', hi, r0),
           ('source1_off', z0, r_off), ('interior_anchor_plane', z_mixed, r0[::-1].copy())]
    return pts


def run(name, bb_source, seed=1234):
    t0 = time.time()
    synth, lf = reference_likelihood(name, seed, bb_source)
    print('%s: reference prepare() %.0f s' % (name, time.time() - t0), flush=True)
    case = dict(config=name, seed=seed, bb_source=bb_source, S=synth.S, n_anchor=list(synth.n_anchor),
                bins=list(synth.bins), calls=[])
    a0 = synth.anchor_ps(0)
    case['checks'] = dict(ps_anchor0_first=[float(v) for v in a0[0, :3]], ps_anchor0_sum=float(a0.sum()),
                          mus_anchor0=[float(v) for v in synth.anchor_mus(0)])
    if bb_source >= 0:
        case['checks']['n_model_anchor0_sum'] = float(synth.anchor_n_model(0).sum())
    for dense in (False, True):
        counts = synth.counts(dense=dense)
        lf.set_data(events_of(synth, counts))
        binned = np.asarray(lf.data_events_per_bin.histogram, dtype=float).ravel()
        assert np.array_equal(binned, counts), 'set_data did not reproduce the counts'
        key = 'dense' if dense else 'sparse'
        case['checks']['counts_%s_sum' % key] = float(counts.sum())
        case['checks']['counts_%s_nonzero' % key] = int(np.count_nonzero(counts))
        for label, z, r in points_of(synth):
            kw = {'shape%d' % i: float(v) for i, v in enumerate(z)}
            kw.update({'s%d_rate_multiplier' % s: float(v) for s, v in enumerate(r)})
            t = time.time()
            try:
                ll = float(lf(**kw))
                asserted = False
            except AssertionError:
                ll, asserted = float('nan'), True
            case['calls'].append(dict(data=key, label=label, z=[float(v) for v in z], mult=[float(v) for v in r],
                                      ll=ll, reference_asserted=asserted))
            print('  %-6s %-22s ll = %r   (%.2f s)' % (key, label, ll, time.time() - t), flush=True)
    return case


if __name__ == '__main__':
    todo = sys.argv[1:] or ['C2', 'C5-2anchor']
    for n in todo:
        assert n in CONFIGS
    out = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            out = json.load(f)
    out['generator'] = dict(reference='blueice %s' % blueice.__version__, numpy=np.__version__, scipy=scipy.__version__,
                            script='tests/golden/make_golden_fullsize.py')
    out.setdefault('cases', {})
    for n in todo:
        out['cases'][n] = run(n, bb_source=0 if n.startswith('C5') else -1)
        with open(OUT, 'w') as f:
            json.dump(out, f, indent=1)
    print('wrote', OUT)
