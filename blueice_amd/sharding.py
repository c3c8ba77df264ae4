"""Sharding independent likelihood evaluations over the GPUs of a node.

The path partitions by evaluation (toy datasets, scan points): one process per GPU, every rank holds a
replica of the anchor tensor, no data-path collective; the only exchange is ONE gather of the fp64 result
vector at the end (`torch.distributed.all_gather` -- RCCL over xGMI with the 'nccl' backend and device
tensors, gloo with host tensors).  A single sequential fit does not shard (replicas only).

The reference has no counterpart: its scans are Python loops over `lf(**kw)`
(blueice/inference.py:49-50,424-432).
"""
import numpy as np

__all__ = ['split_range', 'deal_points_by_cell', 'gather_vector', 'sharded_eval_points', 'sharded_eval_toys',
           'allreduce_sum', 'bin_sharded_eval']


def split_range(n, rank, world):
    """Contiguous, balanced [start, stop) of n units for `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(int(n), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def cell_ids(anchor_z, z):
    """Linear grid-cell id of every point (scipy interval rule: last interval closed); -1 outside the box."""
    z = np.atleast_2d(np.asarray(z, dtype=float))
    ids = np.zeros(len(z), dtype=np.int64)
    ok = np.ones(len(z), dtype=bool)
    for i, g in enumerate(anchor_z):
        g = np.asarray(g, dtype=float)
        n_cells = max(len(g) - 1, 1)
        col = z[:, i]
        ok &= (col >= g[0]) & (col <= g[-1])
        k = np.clip(np.searchsorted(g, col, side='right') - 1, 0, n_cells - 1)
        ids = ids * n_cells + k
    ids[~ok] = -1
    return ids


def deal_points_by_cell(anchor_z, z, world):
    """Assign points to ranks so that points of one grid cell stay together (their corner templates are
    then streamed once per rank) while the loads stay balanced: cell groups, largest first, go to the
    least-loaded rank; a group larger than the fair share is split.  Deterministic, so every rank computes
    the same assignment from the same point list.  -> list of index arrays, one per rank."""
    z = np.atleast_2d(np.asarray(z, dtype=float))
    P = len(z)
    if not len(anchor_z):
        return [np.arange(*split_range(P, r, world)) for r in range(world)]
    ids = cell_ids(anchor_z, z)
    order = np.argsort(ids, kind='stable')
    bounds = np.flatnonzero(np.diff(ids[order])) + 1
    groups = np.split(order, bounds)
    fair = -(-P // world)
    pieces = []
    for g in groups:
        for s in range(0, len(g), fair):
            pieces.append(g[s:s + fair])
    pieces.sort(key=lambda a: (-len(a), int(a[0]) if len(a) else 0))
    load = [0] * world
    mine = [[] for _ in range(world)]
    for piece in pieces:
        r = min(range(world), key=lambda q: (load[q], q))
        mine[r].append(piece)
        load[r] += len(piece)
    return [np.concatenate(m) if m else np.zeros(0, dtype=np.int64) for m in mine]


def gather_vector(local, counts, dist=None, device=None):
    """all_gather of per-rank fp64 vectors of (known) unequal lengths `counts` -> list of numpy arrays.
    `dist` = torch.distributed (already initialised) or None for a single process."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    import torch
    world = dist.get_world_size()
    n_max = max(int(c) for c in counts)
    dev = device if device is not None else ('cuda' if dist.get_backend() == 'nccl' else 'cpu')
    buf = torch.zeros(n_max, dtype=torch.float64, device=dev)
    buf[:len(local)] = torch.from_numpy(local).to(dev)
    parts = [torch.empty(n_max, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(parts, buf)                                # the one collective of the path
    return [p[:int(c)].cpu().numpy() for p, c in zip(parts, counts)]


def sharded_eval_points(eval_fn, anchor_z, z, rate_scale, dist=None):
    """Evaluate P points across the ranks of `dist` and return the full ll [P] on every rank.
    eval_fn(z_local [n, d], rate_local [n, S]) -> ll [n]  (e.g. `lambda z, r: ctx.eval(z, r)[0]`)."""
    z = np.atleast_2d(np.asarray(z, dtype=float))
    rate_scale = np.atleast_2d(np.asarray(rate_scale, dtype=float))
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    deal = deal_points_by_cell(anchor_z, z, world)
    mine = deal[rank]
    local = np.asarray(eval_fn(z[mine], rate_scale[mine]), dtype=np.float64) if len(mine) else np.zeros(0)
    parts = gather_vector(local, [len(d) for d in deal], dist)
    out = np.empty(len(z))
    for idx, vals in zip(deal, parts):
        out[idx] = vals
    return out


def sharded_eval_toys(eval_range_fn, T, dist=None):
    """Toy-MC form: datasets [0, T) split contiguously over ranks; eval_range_fn(t0, t1) -> ll [t1 - t0]
    for the datasets this rank holds.  Returns ll [T] on every rank."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    ranges = [split_range(T, r, world) for r in range(world)]
    t0, t1 = ranges[rank]
    local = np.asarray(eval_range_fn(t0, t1), dtype=np.float64) if t1 > t0 else np.zeros(0)
    return np.concatenate(gather_vector(local, [b - a for a, b in ranges], dist))


def allreduce_sum(local, dist=None):
    """Element-wise sum of an fp64 vector over the ranks (identity for a single process)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.from_numpy(local.copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def bin_sharded_eval(ctx, z, rate_scale, dist=None):
    """The one configuration with a real exchange step: an anchor tensor too large for
    one GPU is sharded over the BIN axis -- `ctx` holds this rank's slice of the bins of every template row and
    of the data -- every rank evaluates all P points on its slice, and the partial log likelihoods are summed
    with ONE all-reduce of P doubles.  Everything in the likelihood is additive over bins; the only global
    quantity, the Beeston-Barlow normalisation sum_b n_model[bb, b], is all-reduced once per model
    (`ctx.bb_totals`) before the first evaluation."""
    if ctx.bb_source >= 0 and not getattr(ctx, '_bb_totals_global', False):
        ctx.bb_totals(allreduce_sum(ctx.bb_totals(), dist))
        ctx._bb_totals_global = True
    ll, status = ctx.eval(z, rate_scale)
    return allreduce_sum(ll, dist), status
