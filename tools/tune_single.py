"""GPU box: the synchronous dense single-point call vs launch shape, three interleaved rounds (the first
measurements of a process run slower: clocks).  python tools/tune_single.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts())
z, r = m.disjoint_cell_points(parity=0, seed=0)
for i in range(3000): ctx.eval_one(z[i % 8], r[i % 8])          # warm the clocks
for rnd in range(2):
    for fmb in (1 << 20,):
        for sb in (2, 3, 4, 5, 6, 7, 8):
            ctx.set_param('fuse_max_blocks', fmb)
            ctx.set_param('single_blocks_per_cu', sb)
            for i in range(50): ctx.eval_one(z[i % 8], r[i % 8])
            ctx.set_param('single_timing_reset', 1)
            t = time.perf_counter()
            for i in range(400): ctx.eval_one(z[i % 8], r[i % 8])
            wall = (time.perf_counter() - t) / 400 * 1e6
            n = ctx.get_param('single_calls')
            sp = [ctx.get_param('single_ns_' + k) / n / 1e3 for k in ('host', 'launch', 'wait')]
            t = time.perf_counter()
            for i in range(400): ctx.eval_one(z[0], r[0] * (1 + 1e-5 * i))
            same = (time.perf_counter() - t) / 400 * 1e6
            print('round %d fuse_max_blocks %7d single_blocks_per_cu %d: rotating cells %.1f us (launch %.1f wait %.1f); same cell %.1f us' % (
                rnd, fmb, sb, wall, sp[1], sp[2], same), flush=True)
ctx.close()
