"""A/B on the GPU box: in-launch finish (sharded tickets) vs a second k_finish launch -- batched headline step and the
synchronous single-point call.  python tools/tune_finish.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
counts = m.counts()
sets = [m.disjoint_cell_points(parity=i, seed=i) for i in range(8)]
for nds in (1, 8):
    ctx.upload_counts(np.stack([m.counts(dataset=i) for i in range(nds)]))
    for fuse in (0, 1):
        ctx.set_param('fuse_finish', fuse)
        plans = [ctx.plan(z, r, dataset=np.arange(8) % nds) for z, r in sets]
        for p in plans: p.run()
        ctx.sync()
        t = time.perf_counter()
        for i in range(200): plans[i % 8].run()
        ctx.sync()
        wall = (time.perf_counter() - t) / 200 * 1e6
        ctx.profile(True)
        for i in range(100): plans[i % 8].run()
        n, ms = ctx.profile_read(); ctx.profile(False)
        print('batched: datasets %d fuse_finish %d: step %.1f us, morph kernel %.1f us (%d launches)' % (nds, fuse, wall, ms / n * 1e3, n), flush=True)
        for p in plans: p.close()
ctx.upload_counts(counts)
z, r = sets[0]
for fmb in (64, 1 << 20):
    ctx.set_param('fuse_max_blocks', fmb)
    for i in range(50): ctx.eval(z[i % 8], r[i % 8])
    ctx.set_param('single_timing_reset', 1)
    t = time.perf_counter()
    for i in range(500): ctx.eval(z[i % 8], r[i % 8])
    wall = (time.perf_counter() - t) / 500 * 1e6
    n = ctx.get_param('single_calls')
    sp = [ctx.get_param('single_ns_' + k) / n / 1e3 for k in ('host', 'launch', 'wait')]
    ctx.profile(True)
    for i in range(64): ctx.eval(z[i % 8], r[i % 8])
    nl, kms = ctx.profile_read(); ctx.profile(False)
    print('single: fuse_max_blocks %d: call %.1f us (host %.1f launch %.1f wait %.1f), first kernel %.1f us' % (fmb, wall, *sp, kms / nl * 1e3), flush=True)
ctx.close()
