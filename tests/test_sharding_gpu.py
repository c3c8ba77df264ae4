"""Two ranks on the GPU box (both on device 0, gloo for the gather -- the box has one GPU): the sharded
scan through the real HIP path equals the single-process device result and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from blueice_amd.device import DeviceContext
    from blueice_amd.sharding import sharded_eval_points, sharded_eval_toys
    from blueice_amd.synthetic import SyntheticModel
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        m = SyntheticModel.named('mini3')
        ctx = DeviceContext(0)
        m.upload(ctx)
        ctx.upload_counts(m.counts(dense=True))
        z, r = m.random_points(301, seed=4)
        ll = sharded_eval_points(lambda zz, rr: ctx.eval(zz, rr)[0], m.anchor_z, z, r, dist)
        toys = np.stack([m.counts(dense=True, dataset=t) for t in range(9)])
        ctx.upload_counts(toys)          # every rank holds all toys here; it evaluates only its range
        z0, r0 = m.default_point()
        lt = sharded_eval_toys(lambda a, b: ctx.eval_datasets(z0, r0, a, b)[0], 9, dist)
        ctx.close()
        q.put((rank, ll, lt))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_gpu_equal_single_process():
    import torch.multiprocessing as mp
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    counts = m.counts(dense=True)
    ctx.upload_counts(counts)
    z, r = m.random_points(301, seed=4)
    single = ctx.eval(z, r)[0]
    dense = m.dense_model()
    want = orc.loglikelihood_batch(dense, counts, z[:12], r[:12])
    toys = np.stack([m.counts(dense=True, dataset=t) for t in range(9)])
    ctx.upload_counts(toys)
    z0, r0 = m.default_point()
    single_t = ctx.eval_datasets(z0, r0)[0]
    ctx.close()
    for rank, ll, lt in got:
        np.testing.assert_allclose(ll, single, rtol=1e-13)
        np.testing.assert_allclose(ll[:12], want, rtol=1e-10)
        np.testing.assert_allclose(lt, single_t, rtol=1e-13)


_TORCH_SCRIPT = r"""
import sys
import numpy as np
import torch
torch.cuda.set_device(0)                      # bench.py's order: torch initialises the GPU first
buf = torch.full((300,), float('nan'), dtype=torch.float64, device='cuda:0')
torch.cuda.synchronize()
sys.path.insert(0, sys.argv[1])
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('mini3')
ctx = DeviceContext(0)
m.upload(ctx)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(100, seed=8)
want, _ = ctx.eval(z, r)
plan = ctx.plan(z, r)
plan.run(buf.data_ptr() + 8 * 100)            # middle third of the torch tensor
ctx.sync()
got = buf.cpu().numpy()
assert np.all(np.isnan(got[:100])) and np.all(np.isnan(got[200:]))
np.testing.assert_array_equal(got[100:200], want)
print('TORCH_TENSOR_OK')
"""


def test_results_land_in_a_torch_device_tensor(tmp_path):
    """bench.py's N > 1 path hands `bi_run_plan` the data pointer of a torch CUDA tensor so that RCCL can
    gather the results without a host round trip: the library must write straight into foreign device
    memory.  Run in a fresh process, torch first -- exactly bench.py's order of initialisation."""
    import subprocess
    script = tmp_path / 'torch_tensor.py'
    script.write_text(_TORCH_SCRIPT)
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'TORCH_TENSOR_OK' in res.stdout, res.stdout + res.stderr


@pytest.mark.parametrize('bb', [-1, 0])
def test_bin_sharded_partials_add_up(bb):
    """Bin sharding (the fallback for tensors larger than one GPU's HBM): two contexts holding complementary bin
    slices; their partial log likelihoods sum to the full one once the Beeston-Barlow totals are global."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini4bb' if bb >= 0 else 'mini3', bb_source=bb)
    dense = m.dense_model()
    counts = m.counts(dense=True)
    z, r = m.random_points(25, seed=6)
    full = DeviceContext(0)
    full.upload_model(dense['anchor_z'], dense['ps'], dense['mus'], n_model=dense['n_model'], bb_source=bb)
    full.upload_counts(counts)
    want, _ = full.eval(z, r)
    cut = m.B // 3 + 1
    parts, ctxs = [], []
    for sl in (slice(0, cut), slice(cut, m.B)):
        c = DeviceContext(0)
        c.upload_model(dense['anchor_z'], dense['ps'][..., sl], dense['mus'],
                       n_model=None if bb < 0 else dense['n_model'][..., sl], bb_source=bb)
        c.upload_counts(counts[sl])
        ctxs.append(c)
    if bb >= 0:
        tot = sum(c.bb_totals() for c in ctxs)            # what the all-reduce does across ranks
        np.testing.assert_allclose(tot, full.bb_totals(), rtol=1e-14)
        for c in ctxs:
            c.bb_totals(tot)
    for c in ctxs:
        parts.append(c.eval(z, r)[0])
        c.close()
    np.testing.assert_allclose(parts[0] + parts[1], want, rtol=1e-12)
    full.close()
