"""The two toy-MC loops of a blueice analysis, written against the reference's own API -- run as

    PYTHONPATH=. python examples/toy_mc_loops.py

(1) BINNED: 10 000 toy datasets of a model with Monte Carlo templates, each fitted ... in the reference a Python loop
    over `d = lf.base_model.simulate(); lf.set_data(d); lf.bestfit_scipy()`.  Here the toys are drawn and evaluated on
    the device (`simulate_toys` / `eval_toys`); the loop below keeps the reference's shape for a few of them to show
    that `set_data` + fit per toy is cheap too (events binned on the device, non-empty-bin tables rebuilt, ~0.3 ms),
    and `bestfit_toys` fits every toy of an ensemble in one call.
(2) UNBINNED: sources whose pdf is a histogram of their own Monte Carlo; `set_data` scores the events at every anchor
    model on the device (`bi_score_events`), so the loop body is a few hundred microseconds instead of tens of
    milliseconds.
"""
import time

import numpy as np

from blueice_amd import BinnedLogLikelihood, UnbinnedLogLikelihood
from blueice_amd.test_helpers import conf_for_test

np.random.seed(1)
conf = conf_for_test(n_sources=2, mc=True, n_events_for_pdf=int(2e5), events_per_day=400.,
                     analysis_space=[['x', np.linspace(-6, 6, 241)]])
conf['sources'] = [dict(name='signal', sigma=0.6, events_per_day=60.), dict(name='background', sigma=2.5)]

# ---- (1) binned ---------------------------------------------------------------------------------------------
lf = BinnedLogLikelihood(conf)
lf.add_rate_parameter('signal')
lf.add_rate_parameter('background')
lf.add_shape_parameter('mu', (-1., 0., 1.))
t = time.perf_counter()
lf.prepare()                                     # 3 anchor models x 2 sources x 2e5 MC events, histogrammed on the device
print('prepare: %.2f s' % (time.perf_counter() - t))
t = time.perf_counter()
lf.simulate_toys(10000, seed=7)                  # drawn on the device at the default parameter point
ll = lf.eval_toys(mu=0.2, signal_rate_multiplier=1.1)
print('10^4 toys drawn and evaluated at one point: %.3f s; mean ll %.3f' % (time.perf_counter() - t, ll.mean()))
t = time.perf_counter()
fits = []
for _ in range(20):                              # the reference's loop shape, for comparison
    lf.set_data(lf.base_model.simulate())
    fits.append(lf.bestfit_scipy()[0]['signal_rate_multiplier'])
print('20 x (simulate on the host, set_data, bestfit_scipy): %.1f ms each; signal multiplier %.2f +- %.2f' % (
    (time.perf_counter() - t) / 20 * 1e3, np.mean(fits), np.std(fits)))

t = time.perf_counter()
lf.simulate_toys(2000, seed=8)                   # ... and the same loop as ONE call: every toy a problem of the batched engine
best, ll = lf.bestfit_toys()
print('2000 toys drawn on the device and all fitted at once: %.3f s; signal multiplier %.2f +- %.2f' % (
    time.perf_counter() - t, np.mean(best['signal_rate_multiplier']), np.std(best['signal_rate_multiplier'])))

# ---- (2) unbinned -------------------------------------------------------------------------------------------
ulf = UnbinnedLogLikelihood(conf)
ulf.add_rate_parameter('signal')
ulf.add_shape_parameter('mu', (-1., 0., 1.))
ulf.prepare()
t = time.perf_counter()
values = []
for _ in range(200):
    ulf.set_data(ulf.base_model.simulate())      # events scored at every anchor model on the device
    values.append(ulf(mu=0.1))
print('200 x (simulate on the host, set_data, one call) of the unbinned likelihood: %.2f ms each (device scoring: %s)' % (
    (time.perf_counter() - t) / 200 * 1e3, ulf._templates not in (None, False)))
