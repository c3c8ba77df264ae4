import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
z, r = m.random_points(1, seed=3)
for T in (8, 100, 256, 1000):
    toys = np.stack([m.counts(dataset=i % 16, dense=True) for i in range(T)])
    ctx.set_param('sparse', 0)
    ctx.upload_counts(toys)
    v = ctx.eval_datasets(z[0], r[0])[0]
    ctx.profile(True)
    for _ in range(3): ctx.eval_datasets(z[0], r[0])
    _, tms = ctx.profile_read(); ctx.profile(False)
    one, _ = ctx.eval(np.repeat(z, 3, 0), np.repeat(r, 3, 0), dataset=np.array([0, 5, T - 1]))
    print('T=%5d dense fp64 counts: %.0f evals/s in kernels (%.2f TB/s of counts), max rel diff vs point path %.1e' % (
        T, 3 * T / (tms * 1e-3), 3 * T * m.B * 8 / (tms * 1e-3) / 1e12, np.max(np.abs(v[[0, 5, T - 1]] - one) / np.abs(one))), flush=True)
