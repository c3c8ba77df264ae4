"""bi_eval_grad throughput against batch size (C2, non-empty-bin form and every bin visited): the iteration cost of the
batched profile-fit engine.  Batches of >= 512 points of one dataset: the matrix-core path (k_grad_mfma, round 4) against
one work item per point (grad_mfma = 0, round 3).  python tools/profile/grad_throughput.py [out.json]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('grad_mfma_min', 512)      # A/B from 512 points on (the default threshold is 2048)
rows = []
NS = 2 ** m.d * m.S
for sparse in (1, 0):
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(m.counts())
    bins = ctx.get_param('nnz_total') if sparse else m.B
    for P in ((64, 511, 512, 1024, 4096, 16384, 131072) if sparse else (64, 1024, 4096)):
        z, r = m.random_points(P, seed=3)
        for mfma in ((1, 0) if P >= 512 else (1,)):
            ctx.set_param('grad_mfma', mfma)
            before = ctx.get_param('n_grad_mfma_launches')
            got = ctx.eval_grad(z, r)
            used = ctx.get_param('n_grad_mfma_launches') > before
            reps = 3 if P * (1 if sparse else 100) <= 200000 else 1
            ctx.profile(True)
            t = time.perf_counter()
            for _ in range(reps):
                ctx.eval_grad(z, r)
            dt = (time.perf_counter() - t) / reps
            n, ms = ctx.profile_read()
            ctx.profile(False)
            # fp64 FMA work of the two matrix products: 2 * NS flop per bin and point each
            tf = 2 * 2.0 * NS * bins * P / (ms / reps * 1e-3) / 1e12 if ms > 0 else None
            rows.append(dict(sparse=sparse, points=P, path='k_grad_mfma' if used else 'one work item per point', ms_per_call=dt * 1e3,
                             us_per_point=dt / P * 1e6, kernels_ms=ms / reps, matrix_products_TFLOPs=tf if used else None))
            print('sparse=%d P=%6d %-24s: %.2f ms per call (%.3f us per point), kernels %.2f ms%s' % (
                sparse, P, rows[-1]['path'], dt * 1e3, dt / P * 1e6, ms / reps, ', %.1f TFLOP/s in the two products' % tf if used else ''), flush=True)
            if mfma == 0 and P >= 512:
                np.testing.assert_allclose(got[0], ref[0], rtol=1e-12)
                np.testing.assert_allclose(got[1], ref[1], rtol=1e-8, atol=1e-8 * np.abs(ref[1]).max())
            ref = got
        ctx.set_param('grad_mfma', 1)
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(dict(workload='bi_eval_grad on C2 (4 sources, 5^3 anchors, 100^3 bins): value + 7 slopes per point, one call',
                       command='python tools/profile/grad_throughput.py', rows=rows), f, indent=1)
ctx.close()
