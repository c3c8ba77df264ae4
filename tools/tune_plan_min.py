"""One-shot cost (plan + run + read) of the host and the device planner versus batch size."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.upload_counts(m.counts())
for sparse in (1, 0):
    ctx.set_param('sparse', sparse)
    for P in (64, 128, 256, 512):
        zz, rr = m.random_points(P, seed=7)
        out = []
        for dev in (0, 1):
            ctx.set_param('device_plan_min', 1 if dev else 1 << 40)
            ts = []
            for rep in range(4):
                t = time.perf_counter()
                p = ctx.plan(zz, rr); p.run(); v = p.read()[0]; p.close()
                ts.append(time.perf_counter() - t)
            out.append(min(ts))
        print('sparse=%d P=%6d: host planner %8.3f ms  device planner %8.3f ms' % (sparse, P, out[0] * 1e3, out[1] * 1e3), flush=True)
