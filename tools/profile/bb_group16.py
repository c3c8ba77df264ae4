"""The Beeston-Barlow scan of a C5 grid cell with 8 and with 16 points per 5.65 GB pass (bb_max_group)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C5-2anchor', bb_source=0)
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(256, seed=3)
ref = None
for g in (8, 16, 8, 16):
    ctx.set_param('bb_max_group', g)
    p = ctx.plan(z, r)
    p.run(); ctx.sync()
    t = time.perf_counter()
    for _ in range(3): p.run()
    ctx.sync()
    dt = (time.perf_counter() - t) / 3
    out = p.read()[0]
    if ref is None: ref = out
    print('bb_max_group %2d: %d launches, %.2f ms per 256 points = %.0f evaluations/s; max rel diff vs first %.1e' % (
        g, p.launches, dt * 1e3, 256 / dt, np.max(np.abs(out - ref) / np.abs(ref))), flush=True)
    p.close()
