import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
for P in (512, 2048, 16384, 131072, 1048576):
    zz, rr = m.random_points(P, seed=7)
    res = {}
    for mf in (0, 1):
        ctx.set_param('scan_mfma', mf)
        t = time.perf_counter(); p = ctx.plan(zz, rr); tp = time.perf_counter() - t
        p.run(); ctx.sync()
        reps = max(3, 262144 // P)
        t = time.perf_counter()
        for _ in range(reps): p.run()
        ctx.sync(); res[mf] = (reps * P / (time.perf_counter() - t), p.read()[0], tp); p.close()
    print('sparse path P=%7d: vector %10.0f/s   matrix-core %10.0f/s (%.2fx, rel diff %.0e)  planning %.2f ms' % (
        P, res[0][0], res[1][0], res[1][0] / res[0][0], np.max(np.abs(res[1][1] - res[0][1]) / np.abs(res[0][1])), res[1][2] * 1e3), flush=True)
