"""A longer toy-MC ensemble on C2 (draw + fit in chunks): throughput against chunk size and HBM in use afterwards."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.synthetic import SyntheticModel
model = SyntheticModel.named('C2')
lf = model.likelihood()
lf.ctx.set_param('compact_budget', 64 << 30)
fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, model.S)}
lf.toy_mc_fits(64, chunk=64, seed=1, **fixed)
for chunk in (256, 512, 1024):
    t = time.perf_counter()
    best, ll = lf.toy_mc_fits(4096, chunk=chunk, seed=5, **fixed)
    dt = time.perf_counter() - t
    print('4096 toys, chunk %4d: %.2f s = %.0f fits/s; rate %.4f +- %.4f; mean max ll %.2f; parked %d MB' % (
        chunk, dt, 4096 / dt, np.mean(best['s0_rate_multiplier']), np.std(best['s0_rate_multiplier']), ll.mean(),
        lf.ctx.get_param('recycle_cache_bytes') >> 20), flush=True)
