"""Where a batched profile scan spends its time: 1024 hypotheses of shape0 on the C2 model through the reference-style
API (first rate + two shape parameters fitted at each), with a cProfile of the host half.  usage: profile_engine.py [points]"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.synthetic import SyntheticModel
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = SyntheticModel.named('C2')
lf = m.likelihood(device=0)
lf.set_binned_data(m.counts().reshape(m.bins))
fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, m.S)}
grid = np.linspace(-1.9, 1.9, P)
lf.bestfit_batched(points={'shape0': grid[:64]}, **fixed)
for kw in (dict(), dict(multi_start=False)):
    t = time.perf_counter()
    best, ll, info = lf.bestfit_batched(points={'shape0': grid}, return_info=True, **dict(fixed, **kw))
    dt = time.perf_counter() - t
    print('%s: %d points in %.3f s = %.0f points/s; %d device calls, %d evaluations, %d starts, iterations %d, converged %.3f stalled %.3f' % (
        kw or 'default', P, dt, P / dt, info['calls'], info['evaluations'], info.get('starts', 1), info['iterations'],
        info['converged'].mean(), info['stalled'].mean()), flush=True)
    single = ll.copy() if kw else None
    multi = ll.copy() if not kw else multi
print('multi-start gain over single start: max %.3e, points improved by > 1e-6: %d' % (np.max(multi - single), np.sum(multi - single > 1e-6)))
pr = cProfile.Profile()
pr.enable()
lf.bestfit_batched(points={'shape0': grid}, **fixed)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
