"""Probe: one-rank RCCL communicator through blueice_amd.comm (run on the GPU box; NCCL_DEBUG=INFO shows the
bootstrap interface RCCL picks)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.comm import connect
from blueice_amd.device import DeviceContext
ctx = DeviceContext(0)
t = time.time()
comm = connect(ctx, backend='rccl', rank=0, world=1)
print('kind', comm.kind, getattr(comm, 'fallback_reason', ''), '%.1f s' % (time.time() - t), flush=True)
if comm.kind == 'rccl':
    print(comm.all_gather(np.arange(4.0)), comm.all_reduce(np.arange(3.0)))
comm.close()
ctx.close()
