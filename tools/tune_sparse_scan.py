"""GPU box: the default (non-empty-bin) path of a scan -- k_morph_reduce<16> on the compacted rows against the
matrix-core scan kernel on the same rows, by scan size.  python tools/tune_sparse_scan.py"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
for P in (16384, 131072, 1000000):
    z, r = m.random_points(P, seed=11)
    for label, lim, cb, wpc in (('k_morph_reduce<16>', 0, 0, 0), ('k_scan_mfma cb=2, split by occupancy', 1 << 30, 2, 0),
                                ('k_scan_mfma cb=4, split by occupancy', 1 << 30, 4, 0), ('k_scan_mfma cb=2 w16', 1 << 30, 2, 16),
                                ('k_scan_mfma cb=2 w24', 1 << 30, 2, 24), ('k_scan_mfma cb=2 w36', 1 << 30, 2, 36), ('k_scan_mfma cb=2 w60', 1 << 30, 2, 60)):
        ctx.set_param('scan_sparse_max_items', lim)
        ctx.set_param('scan_cb', cb)
        ctx.set_param('scan_waves_per_cu', wpc)
        before = ctx.get_param('n_scan_launches')
        p = ctx.plan(z, r)
        p.run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(3): p.run()
        ctx.sync()
        dt = (time.perf_counter() - t) / 3
        print('%8d points, %-38s (scan launches %d, %d waves per cell): %.2f ms, %.1f M evaluations/s' % (P, label, ctx.get_param('n_scan_launches') - before, ctx.get_param('last_scan_nslots'), dt * 1e3, P / dt / 1e6), flush=True)
        p.close()
ctx.close()
