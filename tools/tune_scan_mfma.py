import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.set_param('sparse', 0)
for dense in (False, True):
    ctx.upload_counts(m.counts(dense=dense))
    for P in (16384, 65536, 131072):
        zz, rr = m.random_points(P, seed=7)
        res = {}
        for mf in (0, 1):
            ctx.set_param('scan_mfma', mf)
            p = ctx.plan(zz, rr); p.run(); ctx.sync()
            t = time.perf_counter()
            for _ in range(2): p.run()
            ctx.sync(); res[mf] = (2 * P / (time.perf_counter() - t), p.read()[0]); p.close()
        err = np.max(np.abs(res[1][1] - res[0][1]) / np.abs(res[0][1]))
        print('dense_data=%d P=%6d: vector kernel %8.0f evals/s   matrix-core scan %8.0f evals/s  (%.2fx)  max rel diff %.1e' % (
            dense, P, res[0][0], res[1][0], res[1][0] / res[0][0], err))
