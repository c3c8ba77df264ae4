"""GPU box: the batched engine in its reference-equivalent mode (multi_start=False: the reference's one starting point)
against the reference's own fits on the golden profiled scans, at the grid points where the reference demonstrably converged
(its default settings and tol = 1e-10 agree to 1e-6)."""
import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import model_zoo
from golden_util import GOLDEN_DIR
ns = model_zoo.namespace_of('blueice_amd')
cache = {}
for name, (builder, space, fixed) in model_zoo.PROFILE_SCANS.items():
    f = np.load(os.path.join(GOLDEN_DIR, 'profile_%s.npz' % name))
    lf = cache.setdefault(builder, builder(ns))
    names = [n for n, _ in space]
    grids = np.meshgrid(*[np.asarray(v, dtype=float) for _, v in space], indexing='ij')
    pts = {n: g.ravel() for n, g in zip(names, grids)}
    ref, ref_default = f['ll'].ravel(), f['ll_default'].ravel()
    scale = np.maximum(1.0, np.abs(ref_default))
    conv = np.isfinite(ref) & (np.abs(ref - ref_default) <= 1e-6 * scale)
    for ms in (False, True):
        best, ll, info = lf.bestfit_batched(points=pts, return_info=True, multi_start=ms, **fixed)
        d = ll - ref
        print('%-26s multi_start=%-5s points %4d, reference converged at %4d; there: equal to 1e-6 at %4d, engine higher at %4d (max %+.2e), lower at %d (min %+.2e)' % (
            name, ms, ll.size, conv.sum(), (np.abs(d[conv]) <= 1e-6 * scale[conv]).sum(), (d[conv] > 1e-6 * scale[conv]).sum(),
            d[conv].max() if conv.any() else 0, (d[conv] < -1e-6 * scale[conv]).sum(), d[conv].min() if conv.any() else 0), flush=True)

# the reference's OWN route through the same drivers: a caller's bestfit_routine = scipy's minimiser point by point on the
# device likelihood (batch_stencil=False: the reference's stream of scalar calls) against the reference's default-setting fits
from blueice_amd import inference
plain = lambda lf, **kw: inference.bestfit_scipy(lf, batch_stencil=False, **kw)
for name in ('d2_rate_160', 'd2_rate_120_interior'):
    builder, space, fixed = model_zoo.PROFILE_SCANS[name]
    f = np.load(os.path.join(GOLDEN_DIR, 'profile_%s.npz' % name))
    lf = cache[builder]
    ref_default = f['ll_default']
    scan = inference.likelihood_ratio_scan(lf, *space, bestfit_routine=plain, **fixed)
    want = np.nanmax(ref_default) - ref_default
    d = np.abs(scan - want)
    print('%-26s sequential bestfit_scipy through likelihood_ratio_scan vs the reference default fits: |diff| <= 1e-6 at %d of %d, <= 1e-4 at %d, max %.2e' % (
        name, (d <= 1e-6).sum(), d.size, (d <= 1e-4).sum(), np.nanmax(d)), flush=True)
