"""GPU box: what one `lf.set_data(events)` costs on the C2 model (10^4 events binned on the device, then the data-dependent
tables of the non-empty-bin form), and one fit after it -- the body of a toy-MC loop over host-generated datasets.
python tools/profile/set_data_latency.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.synthetic import SyntheticModel
model = SyntheticModel.named('C2')
lf = model.likelihood(device=0)
rng = np.random.default_rng(5)
names = [n for n, _ in lf.base_model.config['analysis_space']]
def events(n):
    d = np.zeros(n, dtype=[(nm, float) for nm in names] + [('source', int)])
    for nm, (_, e) in zip(names, lf.base_model.config['analysis_space']):
        d[nm] = rng.uniform(e[0], e[-1], n)
    return d
for n in (1000, 10000, 100000):
    sets = [events(n) for _ in range(6)]
    lf.set_data(sets[0]); lf()
    t = time.perf_counter()
    for d in sets[1:]:
        lf.set_data(d)
    dt = (time.perf_counter() - t) / 5
    t = time.perf_counter()
    for d in sets[1:]:
        lf.set_data(d); lf()
    dt2 = (time.perf_counter() - t) / 5
    print('%6d events: set_data %.2f ms, set_data + first call %.2f ms' % (n, dt * 1e3, dt2 * 1e3), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); lf.set_data(sets[2]); lf(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
