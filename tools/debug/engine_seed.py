import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from blueice_amd import profile
from blueice_amd.synthetic import SyntheticModel
seed = 1557
for engine in ('native', 'numpy'):
    profile.ENGINE = engine
    rng = np.random.default_rng(1000 + seed)
    m = SyntheticModel.named('mini3', seed=500 + seed)
    lf = m.likelihood()
    truth = {'shape%d' % i: float(rng.choice([g[1], rng.uniform(g[0], g[-1])])) for i, g in enumerate(m.anchor_z)}
    truth['s0_rate_multiplier'] = float(rng.uniform(0.5, 1.5))
    lf.simulate_toys(4, seed=seed, **truth)
    fixed = dict(s2_rate_multiplier=1., s3_rate_multiplier=1.)
    _, ll = lf.bestfit_toys(**fixed)
    best, ll_all, info = lf.bestfit_toys(multi_start='cells', return_info=True, **fixed)
    single = m.likelihood()
    single.set_binned_data(lf.ctx.download_counts(2).reshape(m.bins))
    res, want = single.bestfit_scipy(use_gradient=True, **fixed)
    print(engine, 'default', ll[2], 'cells', ll_all[2], 'scipy', want, {k: round(float(v[2]) if hasattr(v, '__len__') else v, 4) for k, v in best.items()}, {k: round(v, 4) for k, v in res.items()})
