"""GPU box: the matrix-core gradient path on the small model, step by step (debugging aid)."""
import sys, faulthandler
sys.path.insert(0, '.')
faulthandler.enable()
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('mini3')
ctx = DeviceContext(0)
m.upload(ctx)
ctx.set_param('grad_mfma_min', 512)
for sparse, dense in ((0, True), (2, False)):
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(m.counts(dense=dense))
    rng = np.random.default_rng(11)
    for P, reject in ((600, False), (1700, False), (1700, True)):
        zs = np.array([[rng.uniform(g[0], g[-1]) for g in m.anchor_z] for _ in range(P)])
        rs = rng.uniform(0.3, 1.7, size=(P, m.S))
        if reject:
            zs[5, 0] = 3.0
            rs[7, 0] = -0.5
        for slices in (1, 0):
            ctx.set_param('grad_slices', slices)
            print('sparse', sparse, 'P', P, 'reject', reject, 'slices', slices, flush=True)
            ll, gz, gs, st = ctx.eval_grad(zs, rs)
            ctx.set_param('grad_mfma', 0)
            ll0, gz0, gs0, st0 = ctx.eval_grad(zs, rs)
            ctx.set_param('grad_mfma', 1)
            ok = st0 == 0
            print('   max |dll|', np.abs(ll[ok] - ll0[ok]).max(), 'max |dgz|', np.abs(gz[ok] - gz0[ok]).max(), 'max |dgs|', np.abs(gs[ok] - gs0[ok]).max(), flush=True)
ctx.close()
print('done')
