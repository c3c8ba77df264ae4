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
    for P in (1024, 2048, 4096, 8192):
        zz, rr = m.random_points(P, seed=7)
        res = {}
        for mf in (0, 1):
            ctx.set_param('scan_mfma', mf); ctx.set_param('scan_min_items', 1); ctx.set_param('device_plan_min', 1024)
            p = ctx.plan(zz, rr); p.run(); ctx.sync()
            reps = max(2, 65536 // P)
            t = time.perf_counter()
            for _ in range(reps): p.run()
            ctx.sync(); res[mf] = (reps * P / (time.perf_counter() - t), p.read()[0]); p.close()
        print('dense_data=%d P=%6d (%4.0f items per cell): vector %8.0f/s  matrix-core %8.0f/s (%.2fx, %.0e)' % (
            dense, P, P / 64 / 16, res[0][0], res[1][0], res[1][0] / res[0][0],
            np.max(np.abs(res[1][1] - res[0][1]) / np.abs(res[0][1]))), flush=True)
