"""GPU box: one of 8 ranks' part of a 10^6-point default-path scan of C2 (points resident), dealt two ways: a window of the
cell-sorted list (bi_plan_points_resident with share_world = 8: every rank keys and sorts ALL points, its cells stay together) and a
contiguous RANGE of the caller's points (the same call on a slice, share_world = 1: a rank keys and sorts only its own points, and
meets every grid cell).  python tools/probe/scan_deal_by_index.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
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
send = ctx.device_alloc(8 * P)
for world in (8, 4, 2):
    n = P // world
    off = 3 % world * n
    for how in ('window of the sorted list', 'range of the points'):
        for rep in range(reps):
            ctx.sync()
            t0 = time.perf_counter()
            if how.startswith('window'):
                p = ctx.plan_resident(P, bz, br, None, 3 % world, world)
            else:
                p = ctx.plan_resident(n, bz.ptr + off * 8 * m.d, br.ptr + off * 8 * m.S)
            t1 = time.perf_counter()
            p.run(send.ptr); ctx.sync()
            t2 = time.perf_counter()
            st = p.status()
            t3 = time.perf_counter()
            p.close()
        print('world %d, %-26s: plan %.3f ms, run %.3f ms, status %.3f ms, total %.3f ms' % (
            world, how, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3), flush=True)
ctx.close()
