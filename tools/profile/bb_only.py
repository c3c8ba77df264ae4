"""The Beeston-Barlow pass on one grid cell of configs[4] (2^4 anchors, 6 sources, 50^4 bins: 113 stream rows,
5.65 GB per evaluation), one evaluation per launch -- the command the rocprofv3 passes of profiles/rNN_bb_* wrap.
usage: python tools/profile/bb_only.py [launches]"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = SyntheticModel.named('C5-2anchor', bb_source=0)
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(4, seed=2)
plans = [ctx.plan(z[i], r[i]) for i in range(4)]
for p in plans: p.run()
ctx.sync()
ctx.profile(True)
for i in range(n): plans[i % 4].run()
k, ms = ctx.profile_read(); ctx.profile(False)
print('Beeston-Barlow pass: %d launches, %.1f us each by HIP events, %d algorithmic bytes per launch = %.0f GB/s' % (
    k, ms / k * 1e3, plans[0].bytes, plans[0].bytes * k / (ms * 1e-3) / 1e9))
for p in plans: p.close()
ctx.close()
