import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.upload_counts(m.counts())
z1, r1 = m.random_points(1, seed=3)
for sp in (1, 0):
    ctx.set_param('sparse', sp)
    for name, f in (('eval', lambda: ctx.eval(z1, r1)), ('eval_grad', lambda: ctx.eval_grad(z1, r1)),
                    ('eval 2 points', lambda: ctx.eval(np.repeat(z1, 2, 0), np.repeat(r1, 2, 0)))):
        f()
        ts = []
        for _ in range(1000):
            t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
        print('sparse=%d %-14s median %.1f us' % (sp, name, np.median(ts) * 1e6), flush=True)
