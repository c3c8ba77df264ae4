"""Random synthetic models (independent random templates per anchor: a hump per grid cell): how often the engine's maximum
is below scipy's sequential gradient fit of the same toy, for the default starts and for multi_start='cells'."""
import sys
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = {'default': [], 'cells': []}
failed = 0
for seed in range(n):
    rng = np.random.default_rng(1000 + seed)
    m = SyntheticModel.named('mini3', seed=500 + seed)
    lf = m.likelihood()
    truth = {'shape%d' % i: float(rng.choice([g[1], rng.uniform(g[0], g[-1])])) for i, g in enumerate(m.anchor_z)}
    truth['s0_rate_multiplier'] = float(rng.uniform(0.5, 1.5))
    lf.simulate_toys(4, seed=seed, **truth)
    fixed = dict(s2_rate_multiplier=1., s3_rate_multiplier=1.)
    fits = {'default': lf.bestfit_toys(**fixed)[1], 'cells': lf.bestfit_toys(multi_start='cells', **fixed)[1]}
    single = m.likelihood()
    for t in range(4):
        single.set_binned_data(lf.ctx.download_counts(t).reshape(m.bins))
        try:
            _, want = single.bestfit_scipy(use_gradient=True, **fixed)
        except Exception:
            failed += 1
            continue
        for k in rows:
            rows[k].append(fits[k][t] - want)
for k, v in rows.items():
    v = np.array(v)
    print('%-8s engine - scipy over %d toys: below by > 1e-6 in %d (worst %.2f), above by > 1e-6 in %d (best %+.2f), median %+.3f' % (
        k, len(v), np.sum(v < -1e-6), v.min(), np.sum(v > 1e-6), v.max(), np.median(v)))
print('scipy gave up on %d toys' % failed)
