"""simulate_toys and bestfit_toys of C2 against the number of toys per chunk: where a chunk's time goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.synthetic import SyntheticModel
model = SyntheticModel.named('C2')
lf = model.likelihood()
lf.ctx.set_param('compact_budget', 64 << 30)
fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, model.S)}
lf.simulate_toys(64, seed=1); lf.bestfit_toys(**fixed)
for n in (256, 512, 1024, 256):
    for rep in range(2):
        t0 = time.perf_counter()
        lf.simulate_toys(n, seed=5 + rep)
        t1 = time.perf_counter()
        best, ll, info = lf.bestfit_toys(return_info=True, **fixed)
        t2 = time.perf_counter()
        print('%4d toys: simulate %.3f s (%.2f ms per toy), fit %.3f s (%.2f ms per toy; %d calls, %d evaluations, %d iterations)' % (
            n, t1 - t0, (t1 - t0) / n * 1e3, t2 - t1, (t2 - t1) / n * 1e3, info['calls'], info['evaluations'], info['iterations']), flush=True)
