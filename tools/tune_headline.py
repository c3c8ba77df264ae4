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
sets = [m.disjoint_cell_points(parity=i, seed=i) for i in range(8)]
for bpc in (4, 6, 8, 10, 12, 16):
    for chunks in (1, 8):
        ctx.set_param('blocks_per_cu', bpc); ctx.set_param('tile_chunks', chunks)
        plans = [ctx.plan(z, r) for z, r in sets]
        for p in plans: p.run()
        ctx.sync()
        ctx.profile(True)
        for i in range(160): plans[i % 8].run()
        n, ms = ctx.profile_read(); ctx.profile(False)
        print('blocks_per_cu=%2d tile_chunks=%d: %.2f us per launch, %.3f TB/s' % (bpc, chunks, ms / n * 1e3, plans[0].bytes / (ms / n * 1e-3) / 1e12), flush=True)
        for p in plans: p.close()
