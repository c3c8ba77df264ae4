"""Where a 10^6-point default-path scan step goes: plan (H2D of the points + device planning), run (kernels), read-back."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
z, r = m.random_points(1000000, seed=11)
send = ctx.device_alloc(8 * len(z))
for rep in range(4):
    t0 = time.perf_counter()
    p = ctx.plan(z, r)
    t1 = time.perf_counter()
    p.run(send.ptr); st = p.status()
    t2 = time.perf_counter()
    out = send.to_host(np.float64, len(z))
    t3 = time.perf_counter()
    p.close()
    t4 = time.perf_counter()
    print('plan %.2f ms, run+status %.2f ms, read-back %.2f ms, close %.2f ms, total %.2f ms' % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3), flush=True)
