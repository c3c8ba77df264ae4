import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C5-3anchor', bb_source=0)
m.upload(ctx)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(16, seed=2)
for single in (1, 0):
    ctx.set_param('single_kernel', single)
    for bpc in (2, 4, 6, 8, 12, 16):
        ctx.set_param('blocks_per_cu', bpc)
        for nt in (0, 1, 2):
            ctx.set_param('nt_loads', nt)
            p = ctx.plan(z[:1], r[:1])
            p.run(); ctx.sync()
            ctx.profile(True)
            for _ in range(10): p.run()
            n, ms = ctx.profile_read(); ctx.profile(False)
            print('single_kernel=%d blocks_per_cu=%2d nt=%d: %.3f ms  %.2f TB/s' % (single, bpc, nt, ms / 10, p.bytes / (ms / 10 * 1e-3) / 1e12), flush=True)
            p.close()
