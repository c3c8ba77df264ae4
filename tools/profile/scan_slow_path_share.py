import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
base = m.counts()
z, r = m.random_points(1000000, seed=11)
for name, cnt in (('as drawn', base), ('all ones', np.minimum(base, 1.0)), ('all ones, 9920 bins', None), ('as drawn', base)):
    if cnt is None:
        cnt = np.minimum(base, 1.0).copy()
        nz = np.flatnonzero(cnt)
        cnt[nz[9920:]] = 0.0                     # 155 full strips exactly
    ctx.upload_counts(cnt)
    p = ctx.plan(z, r)
    p.run(); ctx.sync()
    ctx.profile(True)
    for _ in range(3): p.run()
    ctx.sync()
    n, ms = ctx.profile_read(); ctx.profile(False)
    vals, cts = np.unique(cnt[cnt > 0], return_counts=True)
    print('%-22s %d bins with data, counts %s: kernels %.2f ms per scan' % (name, int(ctx.get_param('nnz_total')), dict(zip(vals.astype(int).tolist(), cts.tolist())), ms / 3), flush=True)
    p.close()
ctx.close()
