"""bi_eval_grad throughput against batch size (C2, non-empty-bin form and every bin visited): the iteration cost of the
batched profile-fit engine."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
for sparse in (1, 0):
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(m.counts())
    for P in (64, 256, 511, 512, 1024, 4096, 16384, 131072) if sparse else (64, 1024):
        z, r = m.random_points(P, seed=3)
        ctx.eval_grad(z, r)
        ctx.profile(True)
        t = time.perf_counter()
        for _ in range(3):
            ctx.eval_grad(z, r)
        dt = (time.perf_counter() - t) / 3
        n, ms = ctx.profile_read()
        ctx.profile(False)
        print('sparse=%d P=%6d: %.2f ms per call (%.2f us per point), kernels %.2f ms' % (sparse, P, dt * 1e3, dt / P * 1e6, ms / 3), flush=True)
