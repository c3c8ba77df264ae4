"""GPU box: the dense-data scan (k_scan_mfma, a logarithm per matrix element) against strip width and waves per CU.
python tools/tune_scan_dense.py"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(131072, seed=11)
for cb in (2,):
    for wpc in (24, 45, 60, 90, 135):          # dense data gets a third more: 12 .. 60 waves per CU
        ctx.set_param('scan_cb', cb)
        ctx.set_param('scan_waves_per_cu', wpc)
        p = ctx.plan(z, r)
        p.run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(2): p.run()
        ctx.sync()
        dt = (time.perf_counter() - t) / 2
        print('strip of %d bins, %d waves per CU: %.1f ms, %.0f evaluations/s' % (16 * cb, wpc, dt * 1e3, len(z) / dt), flush=True)
        p.close()
ctx.close()
