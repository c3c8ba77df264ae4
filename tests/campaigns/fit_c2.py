"""BASELINE.json configs[1] end to end through the reference's API: build the C2 likelihood from Source plug-ins,
stream it to the device, set data, run inference.bestfit_scipy; check the maximum against the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.synthetic import SyntheticModel

name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
m = SyntheticModel.named(name)
t = time.perf_counter()
lf = m.likelihood()
print('%s: prepare() incl. streaming %d anchor models to the device: %.1f s' % (name, m.A, time.perf_counter() - t), flush=True)
counts = m.counts()
lf.set_binned_data(counts.reshape(m.bins))
# the synthetic sources are the same noise (degenerate rates): float the first rate and all shape parameters
fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, m.S)}
for label, kw in (('bestfit_scipy()', dict(fixed)), ('bestfit_scipy(use_gradient=True)', dict(fixed, use_gradient=True))):
    t = time.perf_counter()
    best, ll = lf.bestfit_scipy(**kw)
    dt = time.perf_counter() - t
    print('%s: %.3f s, max logL %.6f' % (label, dt, ll))
    print('   ', {k: round(float(v), 5) for k, v in best.items()}, flush=True)
t = time.perf_counter()
n = 300
for i in range(n):
    lf(shape0=0.1 + 1e-4 * i, s0_rate_multiplier=1.05)
print('lf(**params): %.1f us per call (python + device)' % ((time.perf_counter() - t) / n * 1e6))
if '--check' in sys.argv:
    from oracle import blueice_oracle as orc
    z = np.array([best['shape%d' % i] for i in range(m.d)])
    r = np.array([best.get('s%d_rate_multiplier' % s, 1.0) for s in range(m.S)])
    want = orc.loglikelihood(m.cell_model(z), counts, z, r)
    print('oracle at the best-fit point: %.6f   (device %.6f, rel diff %.1e)' % (want, ll, abs(want - ll) / abs(want)))
for P in (10**4, 10**5, 10**6):
    pts = dict(shape0=np.random.default_rng(1).uniform(-2, 2, P), shape1=np.random.default_rng(2).uniform(-2, 2, P),
               s0_rate_multiplier=np.random.default_rng(3).uniform(0.8, 1.2, P))
    lf.eval_points(pts)
    t = time.perf_counter()
    ll = lf.eval_points(pts)
    dt = time.perf_counter() - t
    print('lf.eval_points, %7d points: %.1f ms  (%.2f M evaluations/s end to end)' % (P, dt * 1e3, P / dt / 1e6), flush=True)
