"""The 131 072-point dense scan of the C2 model (every bin visited: k_scan_mfma), a few runs -- the command the
rocprofv3 counter passes of profiles/rNN_scan_* wrap.  usage: python tools/profile/scan_only.py [runs] [dense] [binorder]
(dense: data with an event in nearly every bin; binorder: rows in bin order, scan_pow = 0 -- round 2's kernel path)"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dense = len(sys.argv) > 2
binorder = len(sys.argv) > 3
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.set_param('scan_pow', 0 if binorder else 1)
ctx.upload_counts(m.counts(dense=dense))
z, r = m.random_points(131072, seed=11)
p = ctx.plan(z, r)
p.run(); ctx.sync()
t = time.perf_counter()
for _ in range(runs): p.run()
ctx.sync()
dt = (time.perf_counter() - t) / runs
print('scan of %d points, %s data%s: %.1f ms per run, %.0f evaluations/s, %.1f TFLOP/s of fp64 FMA' % (
    len(z), 'dense' if dense else 'sparse', ' (rows in bin order)' if binorder else '', dt * 1e3, len(z) / dt, 2.0 * 32 * m.B * len(z) / dt / 1e12))
p.close(); ctx.close()
