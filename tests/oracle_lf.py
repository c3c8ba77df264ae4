"""A likelihood stand-in for CPU tests of host-side callers (the batched profile-fit engine, the stencil-batching
objective): the surface those callers use -- rate_parameters, shape_parameters, get_bounds, pdf_base_config,
eval_points, __call__, and optionally values_and_gradients -- served by the CPU oracle on a golden fixture's tensors.
Test infrastructure only: nothing in the package imports it."""
from collections import OrderedDict

import numpy as np

from oracle import blueice_oracle as orc


class OracleLikelihood:
    def __init__(self, model, counts, shape_names, analytic=False):
        self.model, self.counts = model, np.asarray(counts, dtype=float)
        self.S = model['mus'].shape[-1]
        self.source_name_list = ['s%d' % s for s in range(self.S)]
        self.rate_parameters = OrderedDict((n, None) for n in self.source_name_list)
        self.shape_parameters = OrderedDict((n, ({float(z): float(z) for z in g}, None, None))
                                            for n, g in zip(shape_names, model['anchor_z']))
        self.pdf_base_config = {n: 0.0 for n in shape_names}
        self.supports_gradient = analytic
        self.n_calls = self.n_batches = 0

    def make_objective(self, **kw):
        from blueice_amd import inference
        return inference.make_objective(self, **kw)

    def get_bounds(self, name):
        if name in self.shape_parameters:
            zs = list(self.shape_parameters[name][0])
            return min(zs), max(zs)
        return 0, float('inf')

    def _arrays(self, points):
        for k in points:                 # as DeviceLogLikelihood._batch_terms: unknown names are errors, not ignored
            if k not in self.shape_parameters and not (k.endswith('_rate_multiplier') and k[:-16] in self.source_name_list):
                from blueice_amd.exceptions import InvalidParameter
                raise InvalidParameter("%s is not a known shape or rate parameter!" % k)
        P = max([np.size(v) for v in points.values()] + [1])
        z = np.stack([np.broadcast_to(np.asarray(points.get(n, self.pdf_base_config[n]), dtype=float), (P,))
                      for n in self.shape_parameters], axis=1) if self.shape_parameters else np.zeros((P, 0))
        r = np.stack([np.broadcast_to(np.asarray(points.get(n + '_rate_multiplier', 1.0), dtype=float), (P,))
                      for n in self.source_name_list], axis=1)
        return z, r

    def eval_points(self, points, livetime_days=None):
        z, r = self._arrays(points)
        self.n_batches += 1
        self.n_calls += len(z)
        return orc.loglikelihood_batch(self.model, self.counts, z, r)

    def __call__(self, **kw):
        return float(self.eval_points({k: np.array([v]) for k, v in kw.items()})[0])

    def values_and_gradients(self, points, livetime_days=None):
        """Analytic gradient of the Poisson likelihood in numpy (what bi_eval_grad returns on the device)."""
        z, r = self._arrays(points)
        ll = self.eval_points(points)
        grads = OrderedDict((n + '_rate_multiplier', np.full(len(z), np.nan)) for n in self.source_name_list)
        grads.update((n, np.full(len(z), np.nan)) for n in self.shape_parameters)
        h = 1e-6
        for p in range(len(z)):
            if not np.isfinite(ll[p]):
                continue
            mus = orc.interpolate(self.model['anchor_z'], self.model['mus'], z[p]) if len(self.shape_parameters) else self.model['mus']
            ps = orc.interpolate(self.model['anchor_z'], self.model['ps'], z[p]) if len(self.shape_parameters) else self.model['ps']
            ps = ps.reshape(self.S, -1)
            mu = (mus * r[p]) @ ps
            n = self.counts.ravel()
            with np.errstate(all='ignore'):
                w = np.where(n > 0, n / mu, 0.0) - 1.0
            for s, name in enumerate(self.source_name_list):
                grads[name + '_rate_multiplier'][p] = mus[s] * (w @ ps[s])
            for i, name in enumerate(self.shape_parameters):      # shape slopes: differences of the oracle INSIDE the
                # grid cell the point belongs to (g[k] <= z < g[k+1], the last one closed: scipy's and the device's
                # convention), so that a point on an anchor gets the slope of the cell above it, as from bi_eval_grad
                grid = np.asarray(self.model['anchor_z'][i], dtype=float)
                k = int(np.clip(np.searchsorted(grid, z[p][i], side='right') - 1, 0, len(grid) - 2))
                lo, hi = grid[k], grid[k + 1]
                zp, zm = z[p].copy(), z[p].copy()
                zp[i] = min(z[p][i] + h, hi)
                zm[i] = max(z[p][i] - h, lo)
                a = orc.loglikelihood(self.model, self.counts, zp, r[p])
                b = orc.loglikelihood(self.model, self.counts, zm, r[p])
                grads[name][p] = (a - b) / (zp[i] - zm[i])
        return ll, grads
