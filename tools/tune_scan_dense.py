import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.set_param('sparse', 0)
for dense in (False, True):
    ctx.upload_counts(m.counts(dense=dense))
    for mg in (8, 16, 32):
        ctx.set_param('max_group', mg)
        zz, rr = m.random_points(8192, seed=7)
        p = ctx.plan(zz, rr); p.run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(2): p.run()
        ctx.sync(); dt = (time.perf_counter() - t) / 2
        print('dense_data=%d max_group=%2d: %8.0f evals/s  (%.1f fp64 issue slots per point-bin at 100%% VALU)' % (
            dense, mg, 8192 / dt, dt * 39.3e12 / (8192 * 1e6)))
        p.close()
