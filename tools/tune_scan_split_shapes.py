import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
for S, n_anchor, bins, P in ((4, (5, 5, 5), (100, 100, 100), 131072), (8, (5, 5), (400, 400), 65536), (3, (4, 4, 4), (60, 60, 60), 65536), (5, (5, 5), (400, 400), 65536)):
    m = SyntheticModel(S, n_anchor, bins)
    m.upload(ctx, threads=8)
    ctx.set_param('sparse', 0)
    ctx.upload_counts(m.counts(dense=True))
    z, r = m.random_points(P, seed=11)
    for wpc in (0, 16, 32, 64, 96, 128):
        ctx.set_param('scan_waves_per_cu', wpc)
        p = ctx.plan(z, r)
        p.run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(3): p.run()
        ctx.sync()
        dt = (time.perf_counter() - t) / 3
        print('S=%d anchors=%s bins=%d: scan_waves_per_cu %3d -> %.2f ms (waves per cell %d)' % (S, n_anchor, m.B, wpc, dt * 1e3, ctx.get_param('last_scan_nslots')), flush=True)
        p.close()
    ctx.set_param('scan_waves_per_cu', 0)
ctx.close()
