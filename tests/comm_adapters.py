"""Test-side adapter: torch.distributed (gloo, CPU) behind the communicator interface of blueice_amd.comm, to show
that the sharding layer is agnostic of the transport.  The package itself never imports torch."""
import numpy as np


class GlooCommunicator:
    kind = 'gloo'

    def __init__(self, rank, world):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        self.rank, self.world = rank, world

    def all_gather(self, local):
        t = self._torch.from_numpy(np.ascontiguousarray(local).copy())
        parts = [self._torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(parts, t)
        return np.stack([p.numpy() for p in parts])

    def all_reduce(self, local, op='sum'):
        if op == 'bor':
            return np.bitwise_or.reduce(self.all_gather(np.asarray(local).astype(np.int64)), axis=0).astype(np.asarray(local).dtype)
        t = self._torch.from_numpy(np.ascontiguousarray(local).copy())
        self._dist.all_reduce(t, op={'sum': self._dist.ReduceOp.SUM, 'max': self._dist.ReduceOp.MAX,
                                     'min': self._dist.ReduceOp.MIN}[op])
        return t.numpy()

    def broadcast_bytes(self, data=None):
        box = [data]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier(self):
        self._dist.barrier()

    def close(self):
        self._dist.destroy_process_group()
