import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
counts = m.counts()
z, r = m.random_points(64, seed=1)
def lat(n=1000):
    for i in range(20): ctx.eval(z[i], r[i])
    t = time.perf_counter()
    for i in range(n): ctx.eval(z[i % 64], r[i % 64])
    return (time.perf_counter() - t) / n * 1e6
for sparse in (0, 1):
    ctx.set_param('sparse', sparse); ctx.upload_counts(counts)
    for sk, fmb, bpc in ((0, 64, 8), (1, 0, 8), (1, 64, 8), (1, 4096, 8), (1, 4096, 2), (1, 4096, 1), (1, 0, 2), (1, 0, 4)):
        ctx.set_param('single_kernel', sk); ctx.set_param('fuse_max_blocks', fmb); ctx.set_param('blocks_per_cu', bpc)
        print('sparse=%d single_kernel=%d fuse_max_blocks=%4d blocks_per_cu=%d: %.1f us/call' % (sparse, sk, fmb, bpc, lat()))
