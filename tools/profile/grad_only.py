"""bi_eval_grad over 131 072 points of C2 (compacted rows, k_grad_mfma) -- the command for counter passes on that kernel.
usage: python tools/profile/grad_only.py [calls] [points]"""
import sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
ctx.set_param('sparse', 1)
ctx.upload_counts(m.counts())
z, r = m.random_points(P, seed=3)
ctx.eval_grad(z, r)
ctx.profile(True)
t = time.perf_counter()
for _ in range(calls):
    ctx.eval_grad(z, r)
dt = (time.perf_counter() - t) / calls
n, ms = ctx.profile_read(); ctx.profile(False)
bins = ctx.get_param('nnz_total')
print('bi_eval_grad of %d points (%d bins with data): %.2f ms per call, kernels %.2f ms, %.1f TFLOP/s over the two products' % (
    P, bins, dt * 1e3, ms / calls, 2 * 2.0 * 32 * bins * P / (ms / calls * 1e-3) / 1e12))
ctx.close()
