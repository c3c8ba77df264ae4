import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
model = SyntheticModel.named('C2')
lf = model.likelihood(device=0)
counts = model.counts()
lf.set_binned_data(counts.reshape(model.bins))
zz, rr = model.random_points(1000000, seed=11)
pts = {'shape%d' % i: zz[:, i] for i in range(zz.shape[1])}
pts.update({'s%d_rate_multiplier' % s: rr[:, s] for s in range(model.S)})
lf.eval_points(pts)
for _ in range(3):
    t = time.perf_counter(); out = lf.eval_points(pts); print('eval_points 1e6: %.1f ms' % ((time.perf_counter() - t) * 1e3))
pr = cProfile.Profile(); pr.enable(); lf.eval_points(pts); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
