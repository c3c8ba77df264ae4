"""The default-path scan of 10^6 points of C2 (non-empty-bin form), compacted rows ordered by count against bin order."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
z, r = m.random_points(1000000, seed=11)
for pow_on in (0, 1):
    ctx.set_param('sparse', 1)
    ctx.set_param('scan_pow', pow_on)
    ctx.upload_counts(m.counts())
    p = ctx.plan(z, r)
    p.run(); ctx.sync()
    ctx.profile(True)
    t = time.perf_counter()
    for _ in range(3): p.run()
    ctx.sync()
    dt = (time.perf_counter() - t) / 3
    n, ms = ctx.profile_read()
    ctx.profile(False)
    ll, st = p.read()
    print('scan_pow=%d compact_sorted=%d: run %.2f ms = %.1f M evaluations/s; kernels %.2f ms per run; checksum %.6f' % (
        pow_on, ctx.get_param('compact_sorted'), dt * 1e3, len(z) / dt / 1e6, ms / 3, float(ll.sum())), flush=True)
    p.close()
