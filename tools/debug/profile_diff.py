"""Where the batched profile fits differ from the reference goldens (development aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import model_zoo
from golden_util import GOLDEN_DIR

ns = model_zoo.namespace_of('blueice_amd')
built = {}
for name, (builder, space, fixed) in model_zoo.PROFILE_SCANS.items():
    f = np.load(os.path.join(GOLDEN_DIR, 'profile_%s.npz' % name))
    lf = built.get(builder) or built.setdefault(builder, builder(ns))
    names = [n for n, _ in space]
    grids = np.meshgrid(*[np.asarray(v, dtype=float) for _, v in space], indexing='ij')
    best, ll, info = lf.bestfit_batched(points={n: g.ravel() for n, g in zip(names, grids)}, return_info=True, **fixed)
    ref, dflt = f['ll'].ravel(), f['ll_default'].ravel()
    d = ll - ref
    print(name, 'iterations', info['iterations'], 'calls', info['calls'], 'converged', info['converged'].sum(), 'stalled', info['stalled'].sum(),
          'failed', info['failed'].sum(), 'max ours-ref', np.nanmax(d), 'min ours-ref', np.nanmin(d))
    bad = np.flatnonzero(np.abs(d) > 1e-6 * np.maximum(1, np.abs(ref)))
    print('  points off by more than 1e-6:', len(bad))
    for j in bad[:12]:
        print('   ', j, {n: float(g.ravel()[j]) for n, g in zip(names, grids)}, 'ours', ll[j], 'ref', ref[j], 'default', dflt[j],
              'x', [float(best[k][j]) for k in best], 'ref x', f['best'].reshape(len(ref), -1)[j], 'conv', info['converged'][j], 'stalled', info['stalled'][j])
