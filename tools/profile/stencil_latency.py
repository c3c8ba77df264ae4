"""Latency of the small same-cell batches a stencil-batched fit issues (dense C2, every bin visited) against single calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
fuse = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx.set_param('fuse_finish', fuse)
print('fuse_finish', fuse)
for sparse in (0, 1):
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(m.counts())
    z0, r0 = m.default_point()
    for P in (1, 2, 5, 8, 16):
        z = np.tile(z0, (P, 1)) + 1e-8 * np.arange(P)[:, None]
        r = np.tile(r0, (P, 1))
        for _ in range(20):
            ctx.eval(z, r)
        t = time.perf_counter()
        for i in range(300):
            ctx.eval(z, r * (1 + 1e-6 * i))
        dt = (time.perf_counter() - t) / 300
        ctx.profile(True)
        for i in range(50):
            ctx.eval(z, r)
        n, ms = ctx.profile_read()
        ctx.profile(False)
        print('sparse=%d P=%2d: %.1f us per call, kernel %.1f us over %.1f launches' % (sparse, P, dt * 1e6, ms / 50 * 1e3, n / 50), flush=True)
