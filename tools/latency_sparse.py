import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
z, r = m.random_points(64, seed=1)
for i in range(20): ctx.eval(z[i], r[i])
t = time.perf_counter()
for i in range(2000): ctx.eval(z[i % 64], r[i % 64])
print('sparse single call: %.1f us' % ((time.perf_counter() - t) / 2000 * 1e6))
