"""GPU box, under rocprofv3 --kernel-trace --stats: a few calls of bi_eval_datasets_points at configs[2]'s size (10^4 toys of C2),
four rate hypotheses in one grid cell per call, then four points in random cells per call.
python tools/profile/toy_points_trace.py [calls] [same|random|both] [points per call]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else 'both'
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
z, r = m.default_point()
ctx.set_param('sparse', 1)
ctx.generate_toys(z, r, 10000, seed=4242)
zs = np.repeat(z[None, :], P, axis=0)
rs = np.repeat(r[None, :], P, axis=0)
rs[:, 0] *= np.linspace(0.5, 2.0, P)
zr, rr = m.random_points(P, seed=100)
for k in range(n):
    if which in ('same', 'both'):
        ctx.eval_datasets_points(zs + 0.001 * (k % 50), rs)
    if which in ('random', 'both'):
        ctx.eval_datasets_points(zr, rr)
ctx.close()
