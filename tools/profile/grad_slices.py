import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
z, r = m.random_points(131072, seed=3)
for sl in (0, 1, 2, 4, 8, 16, 32):
    ctx.set_param('grad_slices', sl)
    ctx.eval_grad(z, r)
    ctx.profile(True)
    for _ in range(3): ctx.eval_grad(z, r)
    n, ms = ctx.profile_read(); ctx.profile(False)
    print('grad_slices %2d: kernels %.2f ms per call' % (sl, ms / 3), flush=True)
ctx.close()
