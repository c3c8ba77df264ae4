"""Quick regression numbers: headline pass, single-call latency, sparse scan, dense scans."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel

ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.upload_counts(m.counts())

def rate(plan, P, reps):
    plan.run(); ctx.sync()
    t = time.perf_counter()
    for _ in range(reps): plan.run()
    ctx.sync()
    return reps * P / (time.perf_counter() - t)

ctx.set_param('sparse', 0)
zz, rr = m.disjoint_cell_points(0, seed=1)
p = ctx.plan(zz, rr); print('headline 8 disjoint cells: %.0f evals/s' % rate(p, len(zz), 200)); p.close()
z1, r1 = m.random_points(1, seed=3)
for sp in (0, 1):
    ctx.set_param('sparse', sp)
    ctx.eval(z1, r1)
    t = time.perf_counter()
    for _ in range(300): ctx.eval(z1, r1)
    print('single call sparse=%d: %.1f us' % (sp, (time.perf_counter() - t) / 300 * 1e6))
ctx.set_param('sparse', 0)
zz, rr = m.random_points(16384, seed=7)
p = ctx.plan(zz, rr); print('dense path scan 16384 (sparse data): %.0f evals/s' % rate(p, 16384, 3)); p.close()
ctx.set_param('sparse', 1)
zz, rr = m.random_points(131072, seed=7)
p = ctx.plan(zz, rr); print('sparse scan 131072: %.0f evals/s' % rate(p, 131072, 10)); p.close()
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts(dense=True))
zz, rr = m.random_points(16384, seed=7)
p = ctx.plan(zz, rr); print('dense path scan 16384 (dense data): %.0f evals/s' % rate(p, 16384, 3)); p.close()
