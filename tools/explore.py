"""Scratch exploration on the GPU box: time the kernels on the C2 synthetic model."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel

ctx = DeviceContext(0)
print(ctx.info())
m = SyntheticModel.named('C2')
t = time.time(); m.upload(ctx); print('upload %.1fs' % (time.time() - t))
ctx.upload_counts(m.counts())
z, r = m.default_point()

def bench(plan, reps, label, evals):
    plan.run(); ctx.sync()
    t = time.time()
    for _ in range(reps):
        plan.run()
    ctx.sync()
    dt = (time.time() - t) / reps
    ctx.profile(True)
    for _ in range(reps):
        plan.run()
    n, ms = ctx.profile_read(); ctx.profile(False)
    print('%-34s wall %8.1f us  kern %8.1f us (%d launches)  %7.2f TB/s (kernel)  %9.0f evals/s' % (
        label, dt * 1e6, ms / reps * 1e3, n // reps, plan.bytes / (ms / reps * 1e-3) / 1e12, evals / dt))

for bpc in (2, 4, 8):
    ctx.set_param('blocks_per_cu', bpc)
    p = ctx.plan(z, r); bench(p, 50, 'single point bpc=%d' % bpc, 1); p.close()
ctx.set_param('blocks_per_cu', 8)
# same cell, G points
for G in (2, 4, 8, 16):
    zz = np.tile(z, (G, 1)) + np.linspace(0, 0.1, G)[:, None]
    rr = np.tile(r, (G, 1))
    p = ctx.plan(zz, rr); bench(p, 30, 'same cell G=%d' % G, G); p.close()
# random points over all cells
for P in (64, 1024, 4096, 16384):
    zz, rr = m.random_points(P)
    for mg in (16,):
        ctx.set_param('max_group', mg)
        p = ctx.plan(zz, rr); bench(p, 3, 'random P=%d maxG=%d' % (P, mg), P); p.close()
# different cell every call
zs, rs = m.random_points(32, seed=7)
plans = [ctx.plan(zs[i], rs[i]) for i in range(32)]
for p in plans: p.run()
ctx.sync(); t = time.time()
for _ in range(4):
    for p in plans: p.run()
ctx.sync(); dt = (time.time() - t) / 128
print('rotating cells single point: %.1f us/eval  %.2f TB/s wall' % (dt * 1e6, plans[0].bytes / dt / 1e12))
# toys
T = 256
cs = np.stack([m.counts(dataset=i) for i in range(T)])
ctx.upload_counts(cs)
t = time.time(); out, st = ctx.eval_datasets(z, r); dt = time.time() - t
print('toys T=%d: %.1f ms (%.0f evals/s incl. host round trip)' % (T, dt * 1e3, T / dt))
ctx.profile(True); out, st = ctx.eval_datasets(z, r); n, ms = ctx.profile_read(); ctx.profile(False)
print('  kernels %.3f ms over %d launches -> %.0f evals/s' % (ms, n, T / (ms * 1e-3)))
ll, _ = ctx.eval(np.tile(z, (4, 1)), np.tile(r, (4, 1)), dataset=[0, 1, 2, 3])
print('  toy parity vs bi_eval:', out[:4] - ll)
