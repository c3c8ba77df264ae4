"""One fit of C2 (first rate + three shape parameters): scipy's minimiser on the device likelihood (the reference's route,
blueice/inference.py:131-178) against the batched engine with a single problem."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.synthetic import SyntheticModel
model = SyntheticModel.named('C2')
lf = model.likelihood()
lf.set_binned_data(model.counts().reshape(model.bins))
fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, model.S)}
for name, call in (('bestfit_scipy', lambda: lf.bestfit_scipy(**fixed)),
                   ('bestfit_scipy(use_gradient=True)', lambda: lf.bestfit_scipy(use_gradient=True, **fixed)),
                   ('bestfit_batched (one problem, 5 starts)', lambda: lf.bestfit_batched(**fixed)),
                   ('bestfit_batched(multi_start=False)', lambda: lf.bestfit_batched(multi_start=False, **fixed))):
    call()
    t = time.perf_counter()
    for _ in range(5):
        best, ll = call()
    dt = (time.perf_counter() - t) / 5
    print('%-42s %6.2f ms  max ll %.6f  %s' % (name, dt * 1e3, float(np.ravel(ll)[0]), {k: round(float(np.ravel(v)[0]), 4) for k, v in best.items()}))
