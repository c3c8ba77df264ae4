"""The default path of a scan (non-empty-bin form: k_scan_mfma over the compacted rows), 10^6 points of C2 -- the command
for counter passes on that kernel.  usage: python tools/profile/sparse_scan_only.py [runs] [points]"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
z, r = m.random_points(P, seed=11)
p = ctx.plan(z, r)
p.run(); ctx.sync()
t = time.perf_counter()
for _ in range(runs): p.run()
ctx.sync()
dt = (time.perf_counter() - t) / runs
print('scan of %d points, non-empty-bin form (%d bins with data): %.2f ms per run, %.1f M evaluations/s' % (
    len(z), ctx.get_param('nnz_total'), dt * 1e3, len(z) / dt / 1e6))
p.close(); ctx.close()
