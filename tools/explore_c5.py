import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
for bb in (0, -1):
    m = SyntheticModel.named('C5-2anchor', bb_source=bb)
    t = time.time(); m.upload(ctx); print('upload %.1fs' % (time.time() - t))
    for dense in (True, False):
        ctx.set_param('sparse', 0)
        ctx.upload_counts(m.counts(dense=dense))
        z, r = m.random_points(16, seed=2)
        for G in (1, 2, 4, 8, 16):
            p = ctx.plan(z[:G], r[:G])
            p.run(); ctx.sync()
            ctx.profile(True)
            for _ in range(5): p.run()
            n, ms = ctx.profile_read(); ctx.profile(False)
            print('bb=%d dense_data=%d G=%2d: %.3f ms/pass  %.2f TB/s  %.0f evals/s' % (bb, dense, G, ms / 5, p.bytes / (ms / 5 * 1e-3) / 1e12, G / (ms / 5 * 1e-3)))
            p.close()
