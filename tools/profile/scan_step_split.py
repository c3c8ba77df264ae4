"""Where a 10^6-point default-path scan step goes, with the points resident in HBM (bi_plan_points_resident): plan (device
planning), run (kernels), status, read-back -- for the whole batch and for the share of one of 8 ranks.
Under rocprofv3 --kernel-trace --stats the per-kernel times of the planner show (python tools/profile/scan_step_split.py 3)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
P = 1000000
z, r = m.random_points(P, seed=11)
bz, br = ctx.device_alloc(z.nbytes), ctx.device_alloc(r.nbytes)
bz.from_host(z); br.from_host(r)
send, full = ctx.device_alloc(8 * P), ctx.device_alloc(8 * P)
for world in (1, 8):
    for rep in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        p = ctx.plan_resident(P, bz, br, None, 3 if world > 1 else 0, world)
        t1 = time.perf_counter()
        p.run(send.ptr); ctx.sync()
        t2 = time.perf_counter()
        st = p.status()
        t3 = time.perf_counter()
        out = send.to_host(np.float64, P if world == 1 else -(-P // world))
        t4 = time.perf_counter()
        p.close()
        t5 = time.perf_counter()
        print('world %d: plan %.3f ms, run %.3f ms, status %.3f ms, read-back %.3f ms, close %.3f ms, total %.3f ms' % (
            world, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3), flush=True)
ctx.close()
