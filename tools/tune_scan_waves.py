"""GPU box: the 131 072-point scan of C2 with every bin visited, against the number of waves the scan kernels spread a
cell's strips over (scan_waves_per_cu) -- sparse data (non-empty-bin pass + k_scan_valid) and dense data (k_scan_mfma).
python tools/tune_scan_waves.py"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
z, r = m.random_points(131072, seed=11)
for dense in (False, True):
    ctx.upload_counts(m.counts(dense=dense))
    for wpc in (0, 24, 48, 96, 144):              # 0 = the planner's own choice
        ctx.set_param('scan_waves_per_cu', wpc)
        p = ctx.plan(z, r)
        p.run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(3): p.run()
        ctx.sync()
        dt = (time.perf_counter() - t) / 3
        print('%s data, scan_waves_per_cu %d: %.1f ms, %.0f evaluations/s   (waves per cell: scan %d, validity pass %d; resident blocks per CU %d)' % (
            'dense' if dense else 'sparse', wpc, dt * 1e3, len(z) / dt, ctx.get_param('last_scan_nslots'), ctx.get_param('last_valid_nslots'), ctx.get_param('last_scan_resident')), flush=True)
        p.close()
ctx.close()
