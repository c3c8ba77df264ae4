"""Create / use / destroy many contexts and plans; device memory in use must return to where it started."""
import sys, ctypes
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
hip = ctypes.CDLL('libamdhip64.so')
def free_mem():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value
m = SyntheticModel.named('mini3')
ctx = DeviceContext(0); ctx.close()
start = free_mem()
for it in range(300):
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.upload_counts(np.stack([m.counts(dataset=i) for i in range(3)]))
    z, r = m.random_points(700, seed=it)
    ctx.eval(z, r)
    ctx.eval(z[:3], r[:3])
    ctx.eval(z[0], r[0])
    ctx.eval_grad(z[:2], r[:2])
    ctx.eval_datasets(z[0], r[0])
    p = ctx.plan(z, r); p.run(); p.read(); p.close()
    ctx.generate_toys(z[0], r[0], 5, seed=it)
    ctx.eval_datasets(z[1], r[1])
    ctx.close()
    if it % 100 == 99:
        print('after %d contexts: free device memory changed by %+d KB' % (it + 1, (free_mem() - start) // 1024), flush=True)

# round 5: contexts that reach the multi-hypothesis toy kernels (second stream + events with toy_points_overlap = 1), the device
# planner's pinned report block and the Beeston-Barlow-free scan kernels on a larger model
big = SyntheticModel(3, (3, 3), (150, 120), seed=5)
rng = np.random.default_rng(1)
counts = np.zeros((100, big.B))
for t in range(100):
    hit = rng.choice(big.B, size=300, replace=False)
    counts[t, hit] = rng.integers(1, 7, size=300)
start = free_mem()
for it in range(100):
    ctx = DeviceContext(0)
    big.upload(ctx)
    ctx.set_param('sparse', 1)
    ctx.upload_counts(counts)
    z, r = big.random_points(2000, seed=it)
    before = ctx.get_param('n_toy_points_passes')
    ctx.set_param('toy_points_overlap', it % 2)
    ctx.eval_datasets_points(z[:20], r[:20])
    assert ctx.get_param('n_toy_points_passes') > before
    ctx.set_param('device_plan_min', 1)
    ctx.eval(z, r, dataset=rng.integers(0, 100, 2000))
    p = ctx.plan(z, r); p.run(); p.status(); p.close()
    ctx.close()
    if it % 50 == 49:
        print('after %d larger contexts: free device memory changed by %+d KB' % (it + 1, (free_mem() - start) // 1024), flush=True)
