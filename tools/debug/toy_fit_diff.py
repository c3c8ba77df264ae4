"""Where a batched toy fit (bestfit_toys) ends against the sequential fit of the same toy: values, gradients at the
sequential optimum through the multi-dataset path, engine diagnostics."""
import sys
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.synthetic import SyntheticModel
model = SyntheticModel.named('C2')
lf = model.likelihood()
fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, model.S)}
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lf.simulate_toys(T, seed=99)
best, ll, info = lf.bestfit_toys(return_info=True, **fixed)
print('converged %.2f stalled %.2f failed %.2f iterations %d calls %d' % (info['converged'].mean(), info['stalled'].mean(), info['failed'].mean(), info['iterations'], info['calls']))
single = model.likelihood()
worst = 0
for t in range(min(T, 12)):
    cnt = lf.ctx.download_counts(t)
    single.set_binned_data(cnt.reshape(model.bins))
    res, want = single.bestfit_scipy(use_gradient=True, **fixed)
    pts = {k: np.array([v]) for k, v in res.items()}
    pts.update(fixed)
    v_multi, g_multi = lf.values_and_gradients(pts, dataset=np.array([t]))
    v_one, g_one = single.value_and_gradient(**dict(res, **fixed))
    gdiff = max(abs(g_multi[k][0] - g_one[k]) for k in g_one)
    print('toy %2d: batched %.6f sequential %.6f diff %+.3e | at the sequential optimum: value multi-one %+.2e, grad diff %.2e | conv %d stall %d | best %s vs %s' % (
        t, ll[t], want, ll[t] - want, v_multi[0] - v_one, gdiff, info['converged'][t], info['stalled'][t],
        {k: round(float(v[t]), 4) for k, v in best.items()}, {k: round(float(v), 4) for k, v in res.items()}))
