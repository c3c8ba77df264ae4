"""GPU box: bi_eval_datasets_points over 32 hypotheses (8 rate hypotheses in each of 4 grid cells) x 10^4 toys of C2, with and without the
second stream (toy_points_overlap).  python tools/probe/toy_points_overlap.py [calls]"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
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
ref = None
for overlap in (0, 1, 0, 1):
    ctx.set_param('toy_points_overlap', overlap)
    ctx.eval_datasets_points(zs, rs, out=out)
    t = time.perf_counter()
    for k in range(n):
        ctx.eval_datasets_points(zs, rs, out=out)
    dt = (time.perf_counter() - t) / n
    if ref is None:
        ref = out.copy()
    print('toy_points_overlap = %d: %.3f ms per call of 32 x 10^4 (%.1f M evaluations/s), bitwise equal to the first run: %s' % (
        overlap, dt * 1e3, 32e4 / dt / 1e6, np.array_equal(ref, out)), flush=True)
ctx.close()
