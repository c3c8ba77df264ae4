import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.upload_counts(m.counts())
z1, r1 = m.random_points(1, seed=3)
for rep in range(2):
    for chunks in (1, 8):
        ctx.set_param('tile_chunks', chunks)
        for sp in (0, 1):
            ctx.set_param('sparse', sp)
            ctx.eval(z1, r1)
            t = time.perf_counter()
            for _ in range(500): ctx.eval(z1, r1)
            print('tile_chunks=%d single call sparse=%d: %.1f us' % (chunks, sp, (time.perf_counter() - t) / 500 * 1e6), flush=True)
