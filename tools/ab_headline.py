import sys, time, os
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.upload_counts(m.counts())
ctx.set_param('sparse', 0)
sets = [m.disjoint_cell_points(parity=i, seed=i) for i in range(8)]
plans = [ctx.plan(z, r) for z, r in sets]
for rep in range(3):
    for p in plans: p.run()
    ctx.sync()
    ctx.profile(True)
    for i in range(240): plans[i % 8].run()
    n, ms = ctx.profile_read(); ctx.profile(False)
    print('%s: %.2f us per launch' % (os.environ.get('BLUEICE_AMD_LIB', 'default lib'), ms / n * 1e3), flush=True)
