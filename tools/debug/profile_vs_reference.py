"""GPU box: the profile curves of the two golden scans with floating shape parameters (d2_rate_160, d2_rate_120_interior), point by
point: the reference's own fits (golden: default minimiser settings `ll_default`, and tol = 1e-10 `ll`), the reference-equivalent route
of the drop-in (scipy's minimiser point by point on the device likelihood, scalar calls: what a caller's bestfit_routine gives) and the
batched engine's default.  Lists every grid point at which two of them differ by more than 1e-6, and which is higher
("Next round" 8).    python tools/debug/profile_vs_reference.py > profiles/rNN_profile_vs_reference.txt"""
import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import model_zoo
from golden_util import GOLDEN_DIR
from blueice_amd import inference
ns = model_zoo.namespace_of('blueice_amd')
cache = {}
plain = lambda lf, **kw: inference.bestfit_scipy(lf, batch_stencil=False, **kw)
for name in ('d2_rate_160', 'd2_rate_120_interior'):
    builder, space, fixed = model_zoo.PROFILE_SCANS[name]
    f = np.load(os.path.join(GOLDEN_DIR, 'profile_%s.npz' % name))
    lf = cache.setdefault(builder, builder(ns))
    pname, grid = space[0][0], np.asarray(space[0][1], dtype=float)
    ref_default, ref_tight = f['ll_default'].ravel(), f['ll'].ravel()
    # the drop-in's two routes, as log likelihoods at every hypothesis
    best, engine, info = lf.bestfit_batched(points={pname: grid}, return_info=True, **fixed)
    seq = np.array([plain(lf, **dict(fixed, **{pname: float(v)}))[1] for v in grid])
    tol = 1e-6 * np.maximum(1.0, np.abs(ref_default))
    print('== %s: %d hypotheses of %s, fixed %s; free parameters fitted at each' % (name, len(grid), pname, fixed))
    print('   reference default vs reference tol=1e-10 : differ at %3d points (tight higher at %d, by up to %.3g)' % (
        (np.abs(ref_tight - ref_default) > tol).sum(), (ref_tight - ref_default > tol).sum(), np.nanmax(ref_tight - ref_default)))
    print('   sequential route vs reference default     : differ at %3d points' % (np.abs(seq - ref_default) > tol).sum())
    print('   engine default   vs reference default     : differ at %3d points (engine higher at %d, lower at %d)' % (
        (np.abs(engine - ref_default) > tol).sum(), (engine - ref_default > tol).sum(), (ref_default - engine > tol).sum()))
    print('   %-10s %-16s %-16s %-16s %-16s  remark' % (pname[:10], 'reference default', 'reference 1e-10', 'sequential route', 'engine default'))
    for j, v in enumerate(grid):
        d_seq, d_eng = seq[j] - ref_default[j], engine[j] - ref_default[j]
        if abs(d_seq) <= tol[j] and abs(d_eng) <= tol[j] and abs(ref_tight[j] - ref_default[j]) <= tol[j]:
            continue
        remark = []
        if abs(d_seq) > tol[j]:
            remark.append('sequential route %s the reference default by %.3g' % ('ABOVE' if d_seq > 0 else 'BELOW', abs(d_seq)))
        if d_eng > tol[j]:
            remark.append('engine above by %.3g' % d_eng)
        if d_eng < -tol[j]:
            remark.append('ENGINE BELOW by %.3g' % -d_eng)
        print('   %-10.5g %-16.9f %-16.9f %-16.9f %-16.9f  %s' % (v, ref_default[j], ref_tight[j], seq[j], engine[j], '; '.join(remark)))
    print()
