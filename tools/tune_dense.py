"""A/B the dense morph+reduce kernel variants on the bench workload (interleaved rounds, one process)."""
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
sets = [m.disjoint_cell_points(parity=i, seed=i) for i in range(8)]
variants = [(nt, bpc) for nt in (0, 1) for bpc in (2, 4, 8, 16)]
plans = {}
for nt, bpc in variants:
    ctx.set_param('blocks_per_cu', bpc)
    plans[(nt, bpc)] = [ctx.plan(z, r) for z, r in sets]
res = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        ctx.set_param('nt_loads', v[0])
        for p in plans[v]: p.run()
        ctx.sync()
        ctx.profile(True)
        for _ in range(3):
            for p in plans[v]: p.run()
        n, ms = ctx.profile_read(); ctx.profile(False)
        res[v].append(ms / n * 1e3)
for v in variants:
    a = np.array(res[v])
    print('nt=%d blocks_per_cu=%2d: median %.1f us  min %.1f us  -> %.2f TB/s (median)' % (v[0], v[1], np.median(a), a.min(), 2.112e9 / (np.median(a) * 1e-6) / 1e12))
