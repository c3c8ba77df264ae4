"""One bi_eval_grad call on C2 (compacted rows, k_grad_mfma) against the number of slices a cell's blocks are cut into
(grad_slices; 0 = the library's rule): kernel time by HIP events and time per call.  python tools/profile/grad_slices.py [points ...]"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
for P in [int(a) for a in sys.argv[1:]] or [131072]:
    z, r = m.random_points(P, seed=3)
    for sl in (0, 1, 2, 4, 8, 16, 32, 64):
        ctx.set_param('grad_slices', sl)
        ctx.eval_grad(z, r)
        ctx.profile(True)
        t = time.perf_counter()
        for _ in range(5): ctx.eval_grad(z, r)
        dt = (time.perf_counter() - t) / 5
        n, ms = ctx.profile_read(); ctx.profile(False)
        print('%6d points, grad_slices %2d: kernels %.2f ms, %.2f ms per call' % (P, sl, ms / 5, dt * 1e3), flush=True)
ctx.close()
