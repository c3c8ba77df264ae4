"""GPU box: the toy-MC call of configs[2] -- 10^4 datasets drawn on the device, one parameter point per call
(k_morph_logmu + k_dataset_dot_csr + k_dataset_finish) -- wall time per call and the kernels' own time by HIP events.
python tools/profile/toy_only.py [calls]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
z, r = m.default_point()
ctx.set_param('sparse', 1)
if len(sys.argv) > 2:
    ctx.set_param('toy_events', int(sys.argv[2]))      # 0: the bin-by-bin generator
if len(sys.argv) > 3:
    ctx.set_param('dot_lanes', int(sys.argv[3]))       # 8 (default) or 16 lanes per (dataset, tile) run
ctx.generate_toys(z, r, 10000, seed=4242)
out = ctx.device_alloc(8 * 10000)
for k in range(3):
    ctx.eval_datasets_device(out.ptr, z + 0.01 * k, r, 0, 10000)
ctx.sync()
ctx.profile(True)
t = time.perf_counter()
for k in range(n):
    ctx.eval_datasets_device(out.ptr, z + 0.001 * (k % 50), r, 0, 10000)
ctx.sync()
dt = (time.perf_counter() - t) / n
launches, ms = ctx.profile_read(); ctx.profile(False)
res = out.to_host(np.float64, 10000)
print('dot_lanes %d; 10^4 toys per call: %.3f ms per call wall (%.1f M evaluations/s), kernels %.3f ms per call by HIP events (%d launches); checksum %.6f' % (
    ctx.get_param('dot_lanes'), dt * 1e3, 1e4 / dt / 1e6, ms / n, launches, float(res.sum())))
out.free(); ctx.close()
