import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.upload_counts(m.counts())
ctx.set_param('sparse', 0)
z1, r1 = m.random_points(1, seed=3)
zz, rr = m.random_points(64, seed=5)
for keep in (0, -1, 24, 28):
    ctx.set_param('keep_rows', keep)
    for _ in range(20): ctx.eval(z1, r1)
    t = time.perf_counter()
    for i in range(400): ctx.eval(z1 + 1e-7 * i, r1)
    same = (time.perf_counter() - t) / 400
    t = time.perf_counter()
    for i in range(400): ctx.eval(zz[i % 64], rr[i % 64])
    rand = (time.perf_counter() - t) / 400
    print('keep_rows=%2d: same cell %.1f us per call, random cells %.1f us per call' % (keep, same * 1e6, rand * 1e6), flush=True)
