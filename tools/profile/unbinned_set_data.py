"""GPU box: `UnbinnedLogLikelihood.set_data` with histogram-pdf sources -- events scored on the device (bi_score_events)
against the host route (Model.score_events anchor by anchor + upload), 4 sources x 5 x 5 anchor models, 1-D analysis space.
python tools/profile/unbinned_set_data.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd import UnbinnedLogLikelihood
from blueice_amd.test_helpers import conf_for_test

def build(on_device):
    np.random.seed(1)
    conf = conf_for_test(n_sources=4, mc=True, n_events_for_pdf=int(1e5), events_per_day=100.,
                         analysis_space=[['x', np.linspace(-8, 8, 401)]])
    lf = UnbinnedLogLikelihood(conf, likelihood_config=dict(device_scoring=on_device))
    lf.add_shape_parameter('mu', (-2., -1., 0., 1., 2.))
    lf.add_shape_parameter('sigma', (0.6, 0.8, 1., 1.5, 2.))
    t = time.perf_counter(); lf.prepare(); print('prepare (100 source templates of 1e5 events): %.2f s' % (time.perf_counter() - t))
    return lf

for on_device in (True, False):
    rng = np.random.default_rng(2)
    lf = build(on_device)
    for n in (100, 2000, 50000):
        sets = []
        for _ in range(6):
            d = np.zeros(n, dtype=[('x', float), ('source', int)]); d['x'] = rng.normal(0, 1.5, n).clip(-7.9, 7.9); sets.append(d)
        lf.set_data(sets[0]); first = lf(mu=0.3, sigma=1.1)
        t = time.perf_counter()
        for d in sets[1:]:
            lf.set_data(d)
        dt = (time.perf_counter() - t) / 5
        print('%s scoring, %6d events: set_data %.2f ms   (ll of the first set %.9f)' % ('device' if on_device else 'host  ', n, dt * 1e3, first), flush=True)
