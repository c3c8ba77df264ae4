"""Sharding independent likelihood evaluations over the GPUs of a node.

The path partitions by evaluation (toy datasets, scan points): one process per GPU, every rank holds a
replica of the anchor tensor, no data-path collective; the only exchange is ONE gather of the fp64 result
vector at the end, through a communicator of blueice_amd.comm (RCCL over xGMI bound directly, or loopback
sockets for CPU rehearsals; anything with the same five methods works -- the tests also plug in
torch.distributed's gloo).  A single sequential fit does not shard (replicas only).

The reference has no counterpart: its scans are Python loops over `lf(**kw)`
(blueice/inference.py:49-50,424-432).
"""
import numpy as np

from ._capi import ST_INTERNAL
from .exceptions import DeviceError

__all__ = ['split_range', 'deal_points_by_cell', 'gather_vector', 'sharded_eval_points', 'sharded_eval_toys', 'sharded_eval_toys_points',
           'sharded_scan_device', 'allreduce_sum', 'bin_sharded_eval']


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
    the same assignment from the same point list.  -> list of index arrays (ascending), one per rank."""
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
    # ascending within a rank: selecting a rank's share (z[idx]) and scattering its results back (out[idx] = ...) then
    # walk memory forwards; the device planner groups by cell again anyway
    return [np.sort(np.concatenate(m)) if m else np.zeros(0, dtype=np.int64) for m in mine]


def _world(comm):
    return (comm.rank, comm.world) if comm is not None else (0, 1)


def gather_vector(local, counts, comm=None):
    """all_gather of per-rank fp64 vectors of (known) unequal lengths `counts` -> list of numpy arrays."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    rank, world = _world(comm)
    if world == 1:
        return [local]
    n_max = max(int(c) for c in counts)
    buf = np.zeros(max(n_max, 1))
    buf[:len(local)] = local
    parts = comm.all_gather(buf)                                   # the one collective of the path
    return [np.array(p[:int(c)]) for p, c in zip(parts, counts)]


def sharded_eval_points(eval_fn, anchor_z, z, rate_scale, comm=None):
    """Evaluate P points across the ranks of `comm` and return the full ll [P] on every rank.
    eval_fn(z_local [n, d], rate_local [n, S]) -> ll [n]  (e.g. `lambda z, r: ctx.eval(z, r)[0]`)."""
    z = np.atleast_2d(np.asarray(z, dtype=float))
    rate_scale = np.atleast_2d(np.asarray(rate_scale, dtype=float))
    rank, world = _world(comm)
    deal = deal_points_by_cell(anchor_z, z, world)
    mine = deal[rank]
    local = np.asarray(eval_fn(z[mine], rate_scale[mine]), dtype=np.float64) if len(mine) else np.zeros(0)
    parts = gather_vector(local, [len(d) for d in deal], comm)
    out = np.empty(len(z))
    for idx, vals in zip(deal, parts):
        out[idx] = vals
    return out


def sharded_scan_device(ctx, z, rate_scale, comm=None, dataset=None):
    """A scan of P points over the GPUs of `comm` with nothing but the final vector leaving HBM.  -> (ll [P] on every
    rank, run) where run() repeats evaluation + gather on the resident plan (for timing) and returns ll again.

    The dealing happens on the device: every rank hands ALL P points to its context, the device planner's
    (cell, dataset) sort is the same on every rank, and rank r evaluates a contiguous, balanced range of that sorted
    list (`DeviceContext.plan_share`) -- cells stay together, a cell is split over at most two ranks, and there is no
    host pass over the points.  The ranks' result vectors (sorted order) are gathered -- RCCL on the context's stream
    between device buffers when `comm` offers all_gather_device, else through the host communicator -- and one kernel
    scatters them into the caller's point order (`EvalPlan.unsort`); the [P] vector crosses PCIe once.
    Batches the device planner refuses -- infinite rate scales of a source that may go negative, Beeston-Barlow points that
    need the host planner's exact totals (bins with U_b == 0 possible) -- are dealt on the host instead
    (deal_points_by_cell), on every rank alike."""
    z = np.atleast_2d(np.asarray(z, dtype=float))
    rate_scale = np.atleast_2d(np.asarray(rate_scale, dtype=float))
    rank, world = _world(comm)
    P = len(z)

    FAILED = 1 << 30                     # a rank's own step raised: carried to the others as a bit, raised by all together

    def agree(word, error=None):
        # every rank takes part in every collective before anybody raises: a rank that bailed out earlier would leave
        # the others waiting in the gather (ncclAllGather has no timeout)
        if world > 1:
            word = int(comm.all_reduce(np.array([word], dtype=np.int64), 'bor')[0])
        if error is not None:
            raise error
        if word & FAILED:
            raise DeviceError("another rank's share of the scan failed: scan discarded")
        if word & ST_INTERNAL:
            raise DeviceError("a rank's launch gave up waiting for a partial sum (BI_ST_INTERNAL): scan discarded")
        return word

    def status_of(plan):
        try:                                 # nobody reads a plan's status array on these routes: look at its OR
            return plan.status()
        except DeviceError:
            return ST_INTERNAL

    plan = None
    if world > 1 and P > 0:
        # The device planner may refuse the batch (infinite rate scales of a source that may go negative are answered on
        # the host: BI_ERR_INVALID) or fail on one rank only (memory): the ranks agree before anybody takes a route, and
        # if any of them has no plan ALL take the host-dealt route below, which answers such points
        # (the library answers a refusal with BI_ERR_INVALID, which DeviceContext raises as ValueError; a failure of this
        #  rank alone is a DeviceError or anything else -- whatever it is, this rank still takes part in the agreement, or
        #  its peers would wait in the all_reduce for ever)
        try:
            plan = ctx.plan_share(z, rate_scale, dataset, rank, world)
            word = 0
        except Exception:
            plan, word = None, FAILED
        if int(comm.all_reduce(np.array([word], dtype=np.int64), 'bor')[0]) & FAILED:
            if plan is not None:
                plan.close()
            plan = None
    if plan is not None:
        stride = -(-P // world)
        send, recv, full = ctx.device_alloc(8 * stride), ctx.device_alloc(8 * stride * world), ctx.device_alloc(8 * P)
        send.from_host(np.zeros(stride))
        on_device = hasattr(comm, 'all_gather_device')

        def run():
            word, error = 0, None
            try:
                plan.run(send.ptr)
                word = status_of(plan)
            except Exception as e:                   # (a HIP error, a stale plan ...: the others must not wait for this rank)
                word, error = FAILED, e
            if on_device:
                comm.all_gather_device(send.ptr, recv.ptr, stride)
            else:
                recv.from_host(comm.all_gather(send.to_host(np.float64, stride)))
            out = None
            if error is None:
                try:
                    plan.unsort(recv.ptr, stride, full.ptr)
                    out = full.to_host(np.float64, P)
                except Exception as e:
                    word, error = word | FAILED, e
            agree(word, error)
            return out

        return run(), run

    deal = deal_points_by_cell(ctx.anchor_z, z, world)
    mine = deal[rank]
    ds_mine = None if dataset is None else np.broadcast_to(np.asarray(dataset, dtype=np.int64), (P,))[mine]
    plan = ctx.plan(z[mine], rate_scale[mine], ds_mine) if len(mine) else None

    def run():
        local, word, error = np.zeros(len(mine)), 0, None
        if plan is not None:
            try:
                plan.run()
                local, st = plan.read()
                word = int(np.bitwise_or.reduce(st)) if len(st) else 0
            except Exception as e:
                local, word, error = np.zeros(len(mine)), FAILED, e
        parts = gather_vector(local, [len(d) for d in deal], comm)
        out = np.empty(len(z))
        for idx, vals in zip(deal, parts):
            out[idx] = vals
        agree(word & (FAILED | ST_INTERNAL), error)
        return out

    return run(), run


def sharded_eval_toys(eval_range_fn, T, comm=None):
    """Toy-MC form: datasets [0, T) split contiguously over ranks; eval_range_fn(t0, t1) -> ll [t1 - t0]
    for the datasets this rank holds.  Returns ll [T] on every rank."""
    rank, world = _world(comm)
    ranges = [split_range(T, r, world) for r in range(world)]
    t0, t1 = ranges[rank]
    local = np.asarray(eval_range_fn(t0, t1), dtype=np.float64) if t1 > t0 else np.zeros(0)
    return np.concatenate(gather_vector(local, [b - a for a, b in ranges], comm))


def sharded_eval_toys_points(eval_points_fn, anchor_z, z, rate_scale, T, comm=None):
    """Toy-MC over several hypotheses: every rank holds ALL T datasets (each draws the whole ensemble: the toys are numbered
    globally, so N ranks hold the same toys) and the P HYPOTHESES are dealt to the ranks by grid cell -- the hypotheses of a
    cell share the pass over its templates and, four at a time, the pass over the datasets' lists (bi_eval_datasets_points), so
    a rank's call stays as efficient as one process's; splitting the DATASETS of one hypothesis instead makes every rank repeat
    the whole log mu pass.  eval_points_fn(z_local [n, d], rate_local [n, S]) -> ll [n, T].  One gather of n_max * T doubles
    per rank.  Returns ll [P, T] on every rank.  (The reference: the double loop of blueice/inference.py:392-443 over
    hypotheses and simulated datasets, blueice/model.py:69-91.)"""
    z = np.atleast_2d(np.asarray(z, dtype=float))
    rate_scale = np.atleast_2d(np.asarray(rate_scale, dtype=float))
    rank, world = _world(comm)
    P, T = len(z), int(T)
    deal = deal_points_by_cell(anchor_z, z, world)
    mine = deal[rank]
    local = np.asarray(eval_points_fn(z[mine], rate_scale[mine]), dtype=np.float64).reshape(len(mine), T) if len(mine) else np.zeros((0, T))
    parts = gather_vector(local.ravel(), [len(d) * T for d in deal], comm)
    out = np.empty((P, T))
    for idx, vals in zip(deal, parts):
        out[idx] = vals.reshape(len(idx), T)
    return out


def allreduce_sum(local, comm=None):
    """Element-wise sum of an fp64 vector over the ranks (identity for a single process)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if comm is None or comm.world == 1:
        return local
    return comm.all_reduce(local, 'sum')


def bin_sharded_eval(ctx, z, rate_scale, comm=None):
    """The one configuration with a real exchange step: an anchor tensor too large for
    one GPU is sharded over the BIN axis -- `ctx` holds this rank's slice of the bins of every template row and
    of the data -- every rank evaluates all P points on its slice, and the partial log likelihoods are summed
    with ONE all-reduce of P doubles.  Everything in the likelihood is additive over bins; the only global
    quantity, the Beeston-Barlow normalisation sum_b n_model[bb, b], is all-reduced once per model
    (`ctx.bb_totals`) before the first evaluation.  The status bits (Beeston-Barlow assertions, unphysical
    rates) are OR-ed over the ranks: a flag raised on any bin slice is a flag of the point."""
    if ctx.bb_source >= 0 and not getattr(ctx, '_bb_totals_global', False):
        ctx.bb_totals(allreduce_sum(ctx.bb_totals(), comm))
        ctx._bb_totals_global = True
    ll, status = ctx.eval(z, rate_scale)
    if comm is not None and comm.world > 1:
        status = comm.all_reduce(status.astype(np.int64), 'bor').astype(np.int32)
    return allreduce_sum(ll, comm), status
