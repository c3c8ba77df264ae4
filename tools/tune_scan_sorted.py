"""GPU box: k_scan_sorted (rows in count order) against the number of waves a cell's strips are spread over
(scan_waves_per_cu; 0 = the planner's own choice): the 131 072-point scan of C2 over dense data (every bin visited) and the
10^6-point scan on the default path (compacted non-empty bins).  python tools/tune_scan_sorted.py"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
for label, sparse, dense, P, wpcs in (('dense data, every bin', 0, True, 131072, (0, 48, 80, 96)),
                                      ('default path', 1, False, 1000000, (0, 4, 6, 8, 12, 16, 20, 24))):
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(m.counts(dense=dense))
    z, r = m.random_points(P, seed=11)
    for wpc, xcd in [(w, 1) for w in wpcs] + [(0, 0), (0, 2), (0, 1)]:
        ctx.set_param('scan_waves_per_cu', wpc)
        ctx.set_param('scan_xcd', xcd)
        p = ctx.plan(z, r)
        p.run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(3): p.run()
        ctx.sync()
        dt = (time.perf_counter() - t) / 3
        print('%s, %d points, scan_waves_per_cu %d, scan_xcd %d: %.2f ms, %.3f M evaluations/s   (waves per cell %d; resident blocks per CU %d)' % (
            label, P, wpc, xcd, dt * 1e3, P / dt / 1e6, ctx.get_param('last_scan_nslots'), ctx.get_param('last_scan_resident')), flush=True)
        p.close()
ctx.close()
