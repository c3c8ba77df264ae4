"""The 131 072-point scan of C2 over DENSE data (an event in nearly every bin), bin-order rows against count-sorted rows:
wall and kernel time.  The command behind profiles/r03_scan_dense_data_*.  usage: scan_dense_data.py [scan_pow]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel

modes = [int(sys.argv[1])] if len(sys.argv) > 1 else [0, 1]
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(131072, seed=11)
for pow_on in modes:
    ctx.set_param('scan_pow', pow_on)
    t = time.perf_counter()
    p = ctx.plan(z, r)
    t_plan = time.perf_counter() - t
    p.run(); ctx.sync()
    ctx.profile(True)
    t = time.perf_counter()
    p.run(); ctx.sync()
    dt = time.perf_counter() - t
    n, ms = ctx.profile_read()
    ctx.profile(False)
    ll, st = p.read()
    print('scan_pow=%d: plan %.1f ms (first: incl. sorting the rows), run %.2f ms = %.0f evaluations/s; kernel %.2f ms over %d launches; '
          'sorted scans %d; checksum %.6f' % (pow_on, t_plan * 1e3, dt * 1e3, len(z) / dt, ms, n, ctx.get_param('n_sorted_scans'), float(ll.sum())), flush=True)
    p.close()
