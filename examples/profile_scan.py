"""A profiled likelihood scan and an upper limit through the reference's API -- run as

    PYTHONPATH=. python examples/profile_scan.py

`plot_likelihood_ratio` / `one_parameter_interval` of the reference fit the nuisance parameters once per hypothesis, one
scalar likelihood call after the other (blueice/inference.py:332-443).  Here every hypothesis is fitted at the same time:
one device call per optimiser iteration returns value and gradient of all fits still running (blueice_amd/profile.py).
"""
import time

import numpy as np

from blueice_amd.synthetic import SyntheticModel

model = SyntheticModel.named('C2')                       # 4 sources, 3 shape parameters x 5 anchors, 100^3 bins
lf = model.likelihood()                                  # a BinnedLogLikelihood with Source plug-ins, prepared
lf.set_binned_data(model.counts().reshape(model.bins))
fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, model.S)}      # s0's rate and the other shapes float

grid = np.linspace(-1.9, 1.9, 512)
t = time.perf_counter()
ratio = lf.likelihood_ratio_scan(('shape0', grid), **fixed)           # -log likelihood ratio, nuisances profiled out
print('%d profiled points in %.2f s; minimum at shape0 = %.4f' % (len(grid), time.perf_counter() - t, grid[np.argmin(ratio)]))

t = time.perf_counter()
limit = lf.one_parameter_interval('s0_rate_multiplier', bound=3.0, kind='upper', confidence_level=0.9, **fixed)
print('90 %% upper limit on s0_rate_multiplier: %.5f (%.2f s)' % (limit, time.perf_counter() - t))

best, ll = lf.bestfit_batched(points={'shape0': grid[::64]}, **fixed)  # the fits themselves: dict of arrays, maxima
print({k: np.round(v, 4) for k, v in best.items()}, np.round(ll, 3))
