"""C3 at full scale: 10^4 toy datasets of the C2 model generated on the device, then evaluated per call."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
z, r = m.default_point()
for T in (256, 10000):
    t = time.perf_counter(); ctx.generate_toys(z, r, T, seed=42); dt = time.perf_counter() - t
    print('T=%d: generate %.3f s (%.0f toys/s), nnz total %d (%.0f per toy), compact=%d' % (
        T, dt, T / dt, ctx.get_param('nnz_total'), ctx.get_param('nnz_total') / T, ctx.get_param('compact_ready')))
    ctx.eval_datasets(z, r)
    t = time.perf_counter(); ll, st = ctx.eval_datasets(z, r); dt = time.perf_counter() - t
    ctx.profile(True); ctx.eval_datasets(z, r); n, ms = ctx.profile_read(); ctx.profile(False)
    print('   eval_datasets: %.3f ms wall (%.0f evals/s), kernels %.3f ms (%.0f evals/s); mean ll %.2f' % (
        dt * 1e3, T / dt, ms, T / (ms * 1e-3), ll.mean()))
    z2 = z + np.array([0.05, -0.05, 0.02])
    t = time.perf_counter(); ll2, _ = ctx.eval_datasets(z2, r); dt = time.perf_counter() - t
    print('   other point: %.3f ms; mean 2*(ll_true - ll_other) = %.3f' % (dt * 1e3, 2 * (ll - ll2).mean()))
