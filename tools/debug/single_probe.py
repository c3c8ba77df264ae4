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
for i in range(3000):
    ctx.eval_one(z[i % 8], r[i % 8])
def run(label):
    for i in range(50): ctx.eval_one(z[i % 8], r[i % 8])
    t = time.perf_counter()
    for i in range(400): ctx.eval_one(z[i % 8], r[i % 8])
    wall = (time.perf_counter() - t) / 400 * 1e6
    ctx.profile(True)
    for i in range(100): ctx.eval_one(z[i % 8], r[i % 8])
    nl, kms = ctx.profile_read(); ctx.profile(False)
    print('%-40s wall %.1f us, kernel %.1f us' % (label, wall, kms / nl * 1e3), flush=True)
for rnd in range(2):
    run('default (fused finish)')
    ctx.set_param('fuse_max_blocks', 0); run('two launches (morph only timed)'); ctx.set_param('fuse_max_blocks', 1 << 20)
    ctx.set_param('nt_loads', 0); run('default cache policy'); ctx.set_param('nt_loads', 2)
    ctx.set_param('tile_chunks', 1); run('plain tile order'); ctx.set_param('tile_chunks', 8)
    ctx.set_param('keep_rows', 0); run('keep_rows 0'); ctx.set_param('keep_rows', -1)
gb = 0.0
import ctypes as C
for bpc in (2, 4, 8):
    for nt in (1, 0):
        v = C.c_double()
        ctx._check(ctx._lib.bi_measure_stream_bandwidth(ctx._h, 1, 32, nt, bpc, 20, C.byref(v)))
        print('stream ceiling 1 item x 32 rows, nt %d, blocks_per_cu %d: %.0f GB/s = %.1f us per 264 MB' % (nt, bpc, v.value, 264e6 / v.value / 1e3), flush=True)
ctx.close()
