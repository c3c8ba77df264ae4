"""The extended unbinned likelihood on the C2 shape with 10^6 events: k_score_locate + k_score_rows (set_data on the device), then
k_morph_reduce<1,false,true,2>, 8 evaluations per launch in disjoint grid cells -- the command the rocprofv3 passes of
profiles/rNN_unbinned_* wrap (bench.py's `unbinned` leg as a stand-alone).  usage: python tools/profile/unbinned_only.py [launches]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
N = 1000000
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
rng = np.random.default_rng(77)
edges = [np.linspace(0.0, 1.0, b + 1) for b in m.bins]
coords = [rng.uniform(0.0, 1.0, N) for _ in m.bins]
uctx = DeviceContext(0)
ctx.score_events(uctx, 'piecewise', edges, coords)
t = time.perf_counter()
ctx.score_events(uctx, 'piecewise', edges, coords)
score_s = time.perf_counter() - t
sets = [m.disjoint_cell_points(parity=i, seed=50 + i) for i in range(4)]
plans = [uctx.plan(zz, rr) for zz, rr in sets]
PPS = len(sets[0][0])
nbytes = PPS * 8 * (2 ** m.d * m.S) * N
for p in plans: p.run()
uctx.sync()
uctx.profile(True)
for i in range(n): plans[i % 4].run()
k, ms = uctx.profile_read(); uctx.profile(False)
print('unbinned pass: %d launches of %d evaluations over %d events, %.1f us each by HIP events, %d algorithmic bytes per launch = %.0f GB/s; '
      'set_data on the device %.1f ms' % (k, PPS, N, ms / k * 1e3, nbytes, nbytes * k / (ms * 1e-3) / 1e9, score_s * 1e3))
for p in plans: p.close()
uctx.close(); ctx.close()
