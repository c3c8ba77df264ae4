"""The 131 072-point scan of C2 with every bin visited over mostly empty data (non-empty-bin pass + validity pass on the matrix
cores, k_scan_valid) against scan_waves_per_cu (0 = the planner's own choice).  python tools/tune_scan_valid.py"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts())
z, r = m.random_points(131072, seed=11)
for wpc in (0, 8, 12, 16, 24, 32, 48, 64, 96):
    ctx.set_param('scan_waves_per_cu', wpc)
    p = ctx.plan(z, r)
    p.run(); ctx.sync()
    t = time.perf_counter()
    for _ in range(3): p.run()
    ctx.sync()
    dt = (time.perf_counter() - t) / 3
    print('every bin visited, sparse data: scan_waves_per_cu %3d -> %.2f ms, %.3f M evaluations/s' % (wpc, dt * 1e3, len(z) / dt / 1e6), flush=True)
    p.close()
ctx.close()
