"""A likelihood scan sharded over the GPUs of one node -- run as

    python -m blueice_amd.launch --nproc 8 examples/sharded_scan.py

One process per GPU; every rank holds the whole anchor tensor, evaluates the scan points of "its" grid cells and the
per-rank result vectors are gathered ONCE, with RCCL bound directly (blueice_amd.comm) between device buffers.
Replaces the reference's Python double loop over lf(**kw) (blueice/inference.py:424-432).  No PyTorch involved.
"""
import numpy as np

from blueice_amd.comm import connect
from blueice_amd.device import DeviceContext, default_device
from blueice_amd.sharding import sharded_scan_device
from blueice_amd.synthetic import SyntheticModel

model = SyntheticModel.named('C2')                 # 4 sources, 5^3 anchor models, 100^3 bins (synthetic templates)
ctx = DeviceContext(default_device())              # LOCAL_RANK's GPU
comm = connect(ctx, backend='rccl')                # falls back to sockets (together, on every rank) if RCCL cannot start
model.upload(ctx, threads=4)
ctx.upload_counts(model.counts())

z, r = model.random_points(1_000_000, seed=1)      # the same list on every rank
ll, rerun = sharded_scan_device(ctx, z, r, comm)   # ll [1e6] on every rank
if comm.rank == 0:
    best = int(np.argmax(ll))
    print('%d ranks (%s gather): best of %d points: ll = %.6f at z = %s' % (comm.world, comm.kind, len(ll), ll[best], z[best]))
comm.close()
ctx.close()
