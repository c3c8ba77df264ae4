import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
for name, bb in (('C2', -1), ('C5-2anchor', 0)):
    m = SyntheticModel.named(name, bb_source=bb)
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_param('sparse', 0)
    ctx.upload_counts(m.counts(dense=True))
    z, r = m.random_points(16, seed=2)
    for chunks in (1, 2, 4, 8, 16, 64):
        ctx.set_param('tile_chunks', chunks)
        out = []
        for pts in ((z[:1], r[:1]), m.disjoint_cell_points(0, seed=1) if name == 'C2' else (z[:4], r[:4])):
            p = ctx.plan(*pts)
            p.run(); ctx.sync()
            ctx.profile(True)
            for _ in range(20): p.run()
            n, ms = ctx.profile_read(); ctx.profile(False)
            out.append('%d pts: %.3f ms %.2f TB/s' % (len(pts[0]), ms / 20, p.bytes / (ms / 20 * 1e-3) / 1e12))
            p.close()
        print('%s tile_chunks=%3d: %s' % (name, chunks, '   '.join(out)), flush=True)
    ctx.close()
