import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts())
def timeit(plans, reps, label, evals):
    out = []
    for nt in (0, 1):
        ctx.set_param('nt_loads', nt)
        for p in plans: p.run()
        ctx.sync(); t = time.perf_counter()
        for _ in range(reps):
            for p in plans: p.run()
        ctx.sync(); out.append(evals * reps / (time.perf_counter() - t))
    print('%-40s nt=0 %10.0f evals/s   nt=1 %10.0f evals/s' % (label, out[0], out[1]))
z, r = m.stratified_points(seed=3)
timeit([ctx.plan(z, r)], 20, 'all 64 cells, one call', 64)
timeit([ctx.plan(z[i], r[i]) for i in range(64)], 5, 'one point per launch, rotating cells', 64)
timeit([ctx.plan(z[0], r[0])], 300, 'one point per launch, same cell', 1)
zz, rr = m.random_points(4096, seed=7)
timeit([ctx.plan(zz, rr)], 3, 'scan 4096 random points', 4096)
zz = np.tile(z[:1], (16, 1)) + np.linspace(0, 0.01, 16)[:, None]
timeit([ctx.plan(zz, np.tile(r[:1], (16, 1)))], 100, '16 points same cell (G=16)', 16)
sets = [m.disjoint_cell_points(parity=i, seed=i) for i in range(8)]
timeit([ctx.plan(a, b) for a, b in sets], 20, 'bench workload (8 disjoint cells/launch)', 64)
