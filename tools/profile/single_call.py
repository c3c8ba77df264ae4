"""GPU box: the synchronous dense single-point call of configs[1] (bi_eval(P = 1): the call `lf(**kw)` makes ~500 times per fit,
blueice/inference.py:153-165), rotating through 8 cells and repeated in one cell: wall time per call, its split, and the kernel's own
time by HIP events -- by launch shape (single_blocks_per_cu).   python tools/profile/single_call.py [calls] [out.json]"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts())
z, r = m.disjoint_cell_points(parity=0, seed=0)
for i in range(3000):
    ctx.eval_one(z[i % 8], r[i % 8])          # warm the clocks
rows = []
bytes_per_eval = 8.0 * (2 ** m.d * m.S + 1) * m.B
for rnd in range(2):
    for sb in (3, 4, 5, 6, 8):
        ctx.set_param('single_blocks_per_cu', sb)
        for i in range(50):
            ctx.eval_one(z[i % 8], r[i % 8])
        ctx.set_param('single_timing_reset', 1)
        t = time.perf_counter()
        for i in range(n):
            ctx.eval_one(z[i % 8], r[i % 8])
        wall = (time.perf_counter() - t) / n * 1e6
        k = ctx.get_param('single_calls')
        sp = [ctx.get_param('single_ns_' + q) / k / 1e3 for q in ('host', 'launch', 'wait')]
        ctx.profile(True)
        for i in range(100):
            ctx.eval_one(z[i % 8], r[i % 8])
        nl, kms = ctx.profile_read()
        ctx.profile(False)
        t = time.perf_counter()
        for i in range(n):
            ctx.eval_one(z[0], r[0] * (1 + 1e-5 * i))
        same = (time.perf_counter() - t) / n * 1e6
        kus = kms / max(nl, 1) * 1e3
        rows.append(dict(round=rnd, single_blocks_per_cu=sb, wall_us=wall, launch_us=sp[1], wait_us=sp[2], kernel_us=kus,
                         kernel_GB_per_s=bytes_per_eval / (kus * 1e-6) / 1e9, same_cell_wall_us=same))
        print('round %d single_blocks_per_cu %d: rotating cells %.1f us (launch %.1f wait %.1f), kernel %.1f us = %.2f TB/s; same cell %.1f us' % (
            rnd, sb, wall, sp[1], sp[2], kus, bytes_per_eval / (kus * 1e-6) / 1e12, same), flush=True)
ctx.close()
if len(sys.argv) > 2:
    with open(sys.argv[2], 'w') as f:
        json.dump(dict(workload='bi_eval(P = 1), dense C2 (264 MB per evaluation), 8 cells in rotation', command='python tools/profile/single_call.py', rows=rows), f, indent=1)
