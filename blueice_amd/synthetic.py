"""Deterministic synthetic anchor grids of the shapes BASELINE.json names.

The hot path only ever sees tensors, so benchmarks and full-size parity tests use synthetic ones:
    anchors per axis (-2,-1,0,1,2)[:n];  ps[a, s, :] = (U[0,1) + 0.05) normalised to sum 1 (strictly
    positive, so mu > 0);  mus[a, s] = 1000 (s+1) (1 + 0.02 z0 - 0.01 z1 + 0.005 z2 ...);
    counts ~ Poisson(sum_s mus_s ps_s) at the central anchor (about 1e4 events at C2 -> ~1 % of the
    bins are non-empty) or a dense variant with ~10 events per bin.
Every anchor block is generated from its own seed, so the 4 GB C2 tensor (or the 187 GB C5 one) never
has to exist on the host: blocks are streamed to the device one anchor at a time, and a CPU checker
can rebuild just the corners it needs.
"""
import itertools

import numpy as np

from .source import Source

__all__ = ['SyntheticModel', 'CONFIGS', 'TemplateSource']

_MU_SLOPES = (0.02, -0.01, 0.005, 0.0025, -0.00125, 0.0006, 0.0003, 0.0001)

CONFIGS = {
    # name: (S, n_anchor per axis, bins)
    'C1': (2, (3,), (40,)),
    'C2': (4, (5, 5, 5), (100, 100, 100)),
    'C5': (6, (5, 5, 5, 5), (50, 50, 50, 50)),
    'C5-3anchor': (6, (3, 3, 3, 3), (50, 50, 50, 50)),
    'C5-2anchor': (6, (2, 2, 2, 2), (50, 50, 50, 50)),   # one grid cell of C5: full bin count, 4.8 GB
    'mini3': (4, (3, 3, 3), (20, 17, 13)),       # ragged bin count (4420, not a tile multiple)
    'mini4bb': (3, (2, 3, 2, 2), (11, 7, 5, 3)),
}


class SyntheticModel:
    def __init__(self, S, n_anchor, bins, seed=1234, bb_source=-1):
        self.S, self.n_anchor, self.bins = int(S), tuple(int(n) for n in n_anchor), tuple(int(b) for b in bins)
        self.d = len(self.n_anchor)
        self.B = int(np.prod(self.bins, dtype=np.int64))
        self.A = int(np.prod(self.n_anchor, dtype=np.int64)) if self.d else 1
        self.seed = int(seed)
        self.bb_source = int(bb_source)
        full = np.array([-2., -1., 0., 1., 2.])
        self.anchor_z = []
        for n in self.n_anchor:
            if n <= 5:
                lo = (5 - n) // 2
                self.anchor_z.append(full[lo:lo + n].copy())
            else:
                self.anchor_z.append(np.linspace(-2., 2., n))

    def __deepcopy__(self, memo):
        return self                 # immutable in effect; configs holding one are deep-copied per anchor model

    @classmethod
    def named(cls, name, **kw):
        S, na, bins = CONFIGS[name]
        return cls(S, na, bins, **kw)

    # -- anchors ---------------------------------------------------------------------------
    def anchor_multi_index(self, a):
        return np.unravel_index(a, self.n_anchor) if self.d else ()

    def anchor_zs(self, a):
        return tuple(float(g[i]) for g, i in zip(self.anchor_z, self.anchor_multi_index(a)))

    def anchor_ps(self, a):
        """[S, B] PMFs of anchor a (C-order anchor index)."""
        rng = np.random.default_rng([self.seed, 1, int(a)])
        p = rng.random((self.S, self.B))
        p += 0.05
        p /= p.sum(axis=1, keepdims=True)
        return p

    def anchor_mus(self, a):
        zs = self.anchor_zs(a)
        f = 1.0 + sum(sl * z for sl, z in zip(_MU_SLOPES, zs))
        return np.array([1000. * (s + 1) * f for s in range(self.S)])

    def anchor_n_model(self, a):
        """[B] Monte-Carlo counts behind the Beeston-Barlow source (>= 1 everywhere)."""
        rng = np.random.default_rng([self.seed, 2, int(a)])
        return 1.0 + rng.poisson(30., self.B).astype(float)

    def central_anchor(self):
        return int(np.ravel_multi_index(tuple(n // 2 for n in self.n_anchor), self.n_anchor)) if self.d else 0

    # -- data ------------------------------------------------------------------------------
    def counts(self, dense=False, dataset=0, scale=None):
        """[B] Poisson counts around the central anchor's expectation (integer-valued floats)."""
        a = self.central_anchor()
        lam = (self.anchor_mus(a)[:, None] * self.anchor_ps(a)).sum(axis=0)
        if scale is None:
            scale = (10.0 * self.B / lam.sum()) if dense else 1.0
        rng = np.random.default_rng([self.seed, 3, int(dataset), int(bool(dense))])
        return rng.poisson(lam * scale).astype(float)

    # -- points ----------------------------------------------------------------------------
    def default_point(self):
        z = np.array((0.3, -1.7, 1.25, 0.6, -0.4, 0.9, 0.1, -1.1)[:self.d])
        for i, g in enumerate(self.anchor_z):
            z[i] = min(max(z[i], g[0]), g[-1])
        r = np.array((1.1, 1.0, 0.7, 1.0, 1.0, 1.0, 0.9, 1.2)[:self.S])
        return z, r

    def random_points(self, P, seed=0):
        rng = np.random.default_rng([self.seed, 4, int(seed)])
        z = np.stack([rng.uniform(g[0], g[-1], size=P) for g in self.anchor_z], axis=1) if self.d \
            else np.zeros((P, 0))
        r = rng.uniform(0.5, 1.5, size=(P, self.S))
        return z, r

    def stratified_points(self, seed=0):
        """One random point strictly inside every grid cell -> (z [n_cells, d], r [n_cells, S])."""
        rng = np.random.default_rng([self.seed, 5, int(seed)])
        cells = list(itertools.product(*[range(max(len(g) - 1, 1)) for g in self.anchor_z]))
        z = np.empty((len(cells), self.d))
        for j, cell in enumerate(cells):
            for i, (g, k) in enumerate(zip(self.anchor_z, cell)):
                z[j, i] = g[k] + rng.uniform(0.05, 0.95) * (g[k + 1] - g[k]) if len(g) > 1 else g[0]
        r = rng.uniform(0.5, 1.5, size=(len(cells), self.S))
        return z, r

    def disjoint_cell_points(self, parity=0, seed=0):
        """One random point in each cell of a set of grid cells that share NO anchor model with each
        other (per axis cells 0, 2, .. or 1, 3, .. chosen by the bits of `parity`), so a batch of them
        re-uses no template bytes between evaluations.  -> (z [n, d], r [n, S])."""
        rng = np.random.default_rng([self.seed, 6, int(seed), int(parity)])
        per_axis = []
        for i, g in enumerate(self.anchor_z):
            n_cells = max(len(g) - 1, 1)
            start = (parity >> i) & 1 if n_cells > 1 else 0
            per_axis.append(list(range(start, n_cells, 2)) or [0])
        cells = list(itertools.product(*per_axis))
        z = np.empty((len(cells), self.d))
        for j, cell in enumerate(cells):
            for i, (g, k) in enumerate(zip(self.anchor_z, cell)):
                z[j, i] = g[k] + rng.uniform(0.05, 0.95) * (g[k + 1] - g[k]) if len(g) > 1 else g[0]
        r = rng.uniform(0.5, 1.5, size=(len(cells), self.S))
        return z, r

    # -- consumers -------------------------------------------------------------------------
    def upload(self, ctx, threads=1):
        """Stream the model to a DeviceContext one anchor at a time (pdf_morphers.py:62-65 loop); with threads > 1
        the anchor blocks are generated by a small pool a few anchors ahead of the upload (numpy's generators
        release the interpreter lock), so nothing of tensor size ever exists on the host."""
        bb = self.bb_source >= 0

        def block(a):
            return self.anchor_ps(a), self.anchor_mus(a), self.anchor_n_model(a) if bb else None

        ctx.begin_model(self.anchor_z, self.S, self.B, bb_source=self.bb_source)
        if threads <= 1 or self.A == 1:
            for a in range(self.A):
                ctx.set_anchor(a, *block(a))
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(int(threads)) as pool:
                pending = [pool.submit(block, a) for a in range(min(int(threads), self.A))]
                nxt = len(pending)
                for a in range(self.A):
                    ps, mus, nm = pending.pop(0).result()
                    if nxt < self.A:
                        pending.append(pool.submit(block, nxt))
                        nxt += 1
                    ctx.set_anchor(a, ps, mus, nm)
        ctx.end_model()

    def cell_model(self, z):
        """Host tensors restricted to the grid cell containing z -- what a CPU checker needs:
        dict(anchor_z, ps [2.., S, B], mus [2.., S], n_model) with 2 (or 1) anchors per axis.
        Interpolating it at z is identical to interpolating the full tensor."""
        ks = []
        for g, zi in zip(self.anchor_z, z):
            n = len(g)
            if n == 1:
                ks.append((0,))
                continue
            k = n - 2 if zi == g[-1] else min(max(int(np.searchsorted(g, zi, side='right')) - 1, 0), n - 2)
            ks.append((k, k + 1))
        shape = tuple(len(k) for k in ks)
        ps = np.empty(shape + (self.S, self.B))
        mus = np.empty(shape + (self.S,))
        nm = np.empty(shape + (self.S, self.B)) if self.bb_source >= 0 else None
        for loc in itertools.product(*[range(n) for n in shape]):
            a = int(np.ravel_multi_index(tuple(k[i] for k, i in zip(ks, loc)), self.n_anchor)) if self.d else 0
            ps[loc] = self.anchor_ps(a)
            mus[loc] = self.anchor_mus(a)
            if nm is not None:
                nm[loc] = 1.0
                nm[loc][self.bb_source] = self.anchor_n_model(a)
        return dict(anchor_z=[g[list(k)] for g, k in zip(self.anchor_z, ks)], ps=ps, mus=mus, n_model=nm)

    def dense_model(self):
        """Full host tensors (small configs only)."""
        shape = self.n_anchor
        ps = np.empty(shape + (self.S, self.B))
        mus = np.empty(shape + (self.S,))
        nm = np.ones(shape + (self.S, self.B)) if self.bb_source >= 0 else None
        for a in range(self.A):
            loc = self.anchor_multi_index(a)
            ps[loc] = self.anchor_ps(a)
            mus[loc] = self.anchor_mus(a)
            if nm is not None:
                nm[loc][self.bb_source] = self.anchor_n_model(a)
        return dict(anchor_z=self.anchor_z, ps=ps, mus=mus, n_model=nm)


# -- the same synthetic model behind the Source / Model / LogLikelihood API -------------------------------

class TemplateSource(Source):
    """Source number `source_index` of a SyntheticModel at the anchor given by the shape-parameter settings in
    its config (`shape_names`): lets a BinnedLogLikelihood of any size be built through the ordinary plug-in
    route (config['sources'] / default_source_class, blueice/model.py:29-33) without a physics model behind it.
    The PMF grid is generated on demand (nothing of template size stays on the host)."""

    def __init__(self, config, *args, **kwargs):
        synth = config['synthetic_model']
        z = [float(config[name]) for name in config['shape_names']]
        index = []
        for g, zi in zip(synth.anchor_z, z):
            hit = np.flatnonzero(g == zi)
            if len(hit) != 1:
                raise ValueError("TemplateSource exists on the anchors only; got %s" % (z,))
            index.append(int(hit[0]))
        self.synth = synth
        self.anchor = int(np.ravel_multi_index(tuple(index), synth.n_anchor)) if synth.d else 0
        self.index = int(config['source_index'])
        config = dict(config, events_per_day=float(synth.anchor_mus(self.anchor)[self.index]), livetime_days=1)
        super().__init__(config, *args, **kwargs)

    def get_pmf_grid(self):
        ps = self.synth.anchor_ps_cached(self.anchor)[self.index].reshape(self.synth.bins)
        if self.synth.bb_source >= 0:
            return ps, self.synth.anchor_n_model(self.anchor).reshape(self.synth.bins)
        return ps, np.full(self.synth.bins, np.inf)

    def simulate(self, n_events):
        raise NotImplementedError("draw toys with BinnedLogLikelihood.simulate_toys (device side)")


def _anchor_ps_cached(self, a):
    """anchor_ps with a one-entry cache: the S sources of one anchor model ask for the same array in turn."""
    if getattr(self, '_ps_cache', (None, None))[0] != a:
        self._ps_cache = (a, self.anchor_ps(a))
    return self._ps_cache[1]


def _likelihood(self, likelihood_config=None, device=0, **kwargs):
    """A prepared blueice_amd.likelihood.BinnedLogLikelihood of this model: one rate parameter per source, one
    shape parameter per axis ('shape0', ...), analysis space = unit-width bins per axis.  No data set yet."""
    from .likelihood import BinnedLogLikelihood
    names = ['shape%d' % i for i in range(self.d)]
    config = dict(
        analysis_space=[('x%d' % i, np.arange(b + 1, dtype=float)) for i, b in enumerate(self.bins)],
        default_source_class=TemplateSource, synthetic_model=self, shape_names=names,
        sources=[dict(name='s%d' % s, source_index=s) for s in range(self.S)],
        **{name: float(g[len(g) // 2]) for name, g in zip(names, self.anchor_z)})
    lc = dict(likelihood_config or {}, device=device)
    if self.bb_source >= 0:
        lc.update(model_statistical_uncertainty_handling='bb_single', bb_single_source=self.bb_source)
    lf = BinnedLogLikelihood(config, likelihood_config=lc, **kwargs)
    for s in range(self.S):
        lf.add_rate_parameter('s%d' % s)
    for name, g in zip(names, self.anchor_z):
        lf.add_shape_parameter(name, tuple(float(v) for v in g))
    lf.prepare()
    return lf


SyntheticModel.anchor_ps_cached = _anchor_ps_cached
SyntheticModel.likelihood = _likelihood
