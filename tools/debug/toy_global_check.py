"""How often a batched toy fit of the 40-bin test model ends on the global maximum: the engine's maxima against a brute-force
grid over the whole shape range (the likelihood of low-statistics toys can have one hump per grid cell of the morph)."""
import sys
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import model_zoo
ns = model_zoo.namespace_of('blueice_amd')
lf = model_zoo.fit_c1_like(ns)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lf.simulate_toys(T, seed=11, s0_rate_multiplier=0.9, shift=0.0)
best, ll, info = lf.bestfit_toys(return_info=True)
print('calls', info['calls'], 'evaluations', info['evaluations'], 'starts', info.get('starts'))
short = []
for t in range(T):
    c0, c1 = best['s0_rate_multiplier'][t], best['s1_rate_multiplier'][t]
    r0 = np.linspace(max(0.0, c0 - 0.6), c0 + 0.6, 49); r1 = np.linspace(max(0.0, c1 - 0.6), c1 + 0.6, 49); sh = np.linspace(-1, 1, 81)
    g0, g1, gs = np.meshgrid(r0, r1, sh, indexing='ij')
    gl = lf.eval_points({'s0_rate_multiplier': g0.ravel(), 's1_rate_multiplier': g1.ravel(), 'shift': gs.ravel()}, dataset=np.full(g0.size, t))
    top = np.nanmax(gl)
    if top > ll[t] + 1e-6:
        short.append((t, round(float(top - ll[t]), 5), round(float(best['shift'][t]), 3), round(float(gs.ravel()[np.nanargmax(gl)]), 3)))
print('%d of %d toys below the grid maximum:' % (len(short), T), short)
