"""GPU box: bi_eval_datasets_points over 32 hypotheses x 10^4 toys of C2 by lanes per (dataset, tile) run of the dot kernel
(toy_points_lanes = 2 / 4 / 8: 48 / 64 / 64 entry slots per run in registers).  python tools/probe/toy_points_lanes.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
z, r = m.default_point()
ctx.set_param('sparse', 1)
ctx.generate_toys(z, r, 10000, seed=4242)
zc, _ = m.random_points(4, seed=5)
zs = np.repeat(zc, 8, axis=0)
rs = np.repeat(r[None, :], 32, axis=0)
rs[:, 0] *= np.tile(np.linspace(0.5, 2.0, 8), 4)
out = np.empty((32, 10000))
for lanes in (0, 2, 4, 8, 2, 4):
    ctx.set_param('toy_points_lanes', lanes)
    ctx.eval_datasets_points(zs, rs, out=out)
    t = time.perf_counter()
    for k in range(n):
        ctx.eval_datasets_points(zs, rs, out=out)
    dt = (time.perf_counter() - t) / n
    print('toy_points_lanes = %d: %.3f ms per call of 32 x 10^4 (%.1f M evaluations/s)' % (lanes, dt * 1e3, 32e4 / dt / 1e6), flush=True)
ctx.close()
