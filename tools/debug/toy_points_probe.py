"""Why a random fuzz model does or does not reach the multi-point toy kernels: prints the quantities the gate reads."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
from blueice_amd.device import DeviceContext
import test_fuzz_gpu as F
rng = np.random.default_rng(17000)
ctx = DeviceContext(0)
for B in (12289, 20001):
    model, counts0 = F.random_case(rng, 2, 4, B, -1)
    T = 97
    counts = np.stack([rng.poisson(counts0 * 3.0).astype(float) for _ in range(T)])
    ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
    ctx.set_param('sparse', 1)
    ctx.set_param('dot_entry16', 1)
    ctx.upload_counts(counts)
    ctx.set_param('toy_points_pp', 0)
    ctx.set_param('toy_points_lanes', 0)
    print('params', {k: ctx.get_param(k) for k in ('dot_entry16', 'toy_points_pp', 'toy_points_lanes', 'dot_tiled', 'tmm_entry_bytes')})
    z, r = F.random_points(rng, model, 5, 4)
    before = ctx.get_param('n_toy_points_passes')
    got, st = ctx.eval_datasets_points(z, r)
    print(B, 'nnz', int((counts > 0).sum()), 'per dataset and tile', (counts > 0).sum() / T / -(-B // 4096),
          'passes', ctx.get_param('n_toy_points_passes') - before, 'status', st, 'sparse', ctx.get_param('sparse'))
ctx.close()
