"""A Beeston-Barlow scan of 256 points in one grid cell of configs[4] (6 sources, 2^4 anchors, 50^4 bins; the C5-BB-scan leg of
bench.py) -- the command for kernel traces and counter passes on k_scan_bb.   python tools/profile/bb_scan_only.py [runs] [points] [scan_bb] [dense data 1|0]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = SyntheticModel.named('C5-2anchor', bb_source=0)
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.set_param('device_plan_min', 1)
if len(sys.argv) > 3:
    ctx.set_param('scan_bb', int(sys.argv[3]))
dense = int(sys.argv[4]) if len(sys.argv) > 4 else 1
counts = m.counts(dense=bool(dense))
ctx.upload_counts(counts)
z, r = m.random_points(P, seed=900)
plan = ctx.plan(z, r)
plan.run(); ctx.sync()
ctx.profile(True)
t = time.perf_counter()
for _ in range(runs):
    plan.run()
ctx.sync()
dt = (time.perf_counter() - t) / runs
n, ms = ctx.profile_read(); ctx.profile(False)
ll, st = plan.read()
# fp64 FMA work of the three products: 2 flop x (2^d (S - 1) + 2 2^d) streams per bin and point
flops = 2.0 * (16 * 5 + 32) * m.B * P
print('%d events in %d bins (%.1f %% of the 16-bin tiles without events); ' % (counts.sum(), m.B, 100.0 * (counts[:m.B // 16 * 16].reshape(-1, 16).sum(axis=1) == 0).mean()), end='')
print('Beeston-Barlow scan of %d points (%s): %.2f ms per run, kernels %.2f ms = %.0f evaluations/s, %.1f TFLOP/s in the three products; status OR %d' % (
    P, 'k_scan_bb' if ctx.get_param('n_bb_scan_launches') else 'k_morph_reduce<8,true>', dt * 1e3, ms / runs, P / (ms / runs * 1e-3),
    flops / (ms / runs * 1e-3) / 1e12, int(np.bitwise_or.reduce(st))))
plan.close()
ctx.close()
