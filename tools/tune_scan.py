import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts())
for P in (1024, 4096, 16384, 50000):
    zz, rr = m.random_points(P, seed=7)
    res = []
    for xa in (0, 1, 0, 1):
        ctx.set_param('xcd_affine', xa)
        p = ctx.plan(zz, rr); p.run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(3): p.run()
        ctx.sync(); res.append(3 * P / (time.perf_counter() - t)); p.close()
    print('P=%6d  xcd_affine=0: %8.0f / %8.0f evals/s   xcd_affine=1: %8.0f / %8.0f evals/s' % (P, res[0], res[2], res[1], res[3]))
z, r = m.stratified_points(seed=3)
for xa in (0, 1):
    ctx.set_param('xcd_affine', xa)
    p = ctx.plan(z, r); p.run(); ctx.sync(); t = time.perf_counter()
    for _ in range(20): p.run()
    ctx.sync(); print('all 64 cells xcd_affine=%d: %.0f evals/s' % (xa, 20 * 64 / (time.perf_counter() - t))); p.close()
