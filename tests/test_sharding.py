"""The N > 1 path on CPU: two ranks shard a scan / a toy-MC batch, evaluate their share (with the CPU oracle
standing in for the device -- tests may use it as the checker) and gather; the result must equal the
single-process evaluation, and the dealing must keep cells together and balanced.  Run over both transports: the
package's own loopback-socket communicator (blueice_amd.comm, also the bootstrap channel of the RCCL one) and
torch.distributed's gloo behind the same interface (tests/comm_adapters.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_range_covers_everything():
    from blueice_amd.sharding import split_range
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [split_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_deal_points_by_cell_balanced_and_cell_local():
    from blueice_amd.sharding import cell_ids, deal_points_by_cell
    rng = np.random.default_rng(0)
    grid = [np.array([-2., -1., 0., 1., 2.])] * 3
    z = rng.uniform(-2, 2, size=(5000, 3))
    z[:5] = [[2, 2, 2], [-2, -2, -2], [0, 0, 0], [3, 0, 0], [np.nan, 0, 0]]
    ids = cell_ids(grid, z)
    assert ids[0] == 63 and ids[1] == 0 and ids[3] == -1 and ids[4] == -1
    for world in (1, 2, 8):
        deal = deal_points_by_cell(grid, z, world)
        allidx = np.sort(np.concatenate(deal))
        np.testing.assert_array_equal(allidx, np.arange(len(z)))
        sizes = [len(d) for d in deal]
        assert max(sizes) - min(sizes) <= 0.05 * len(z) / world + 1
        # a cell is split over at most 2 ranks here (groups are far smaller than the fair share)
        owners = {}
        for r, d in enumerate(deal):
            for c in np.unique(ids[d]):
                owners.setdefault(int(c), set()).add(r)
        assert max(len(v) for v in owners.values()) <= 2
    one = deal_points_by_cell(grid, np.tile([[0.5, 0.5, 0.5]], (100, 1)), 4)    # one hot cell is split
    assert sorted(len(d) for d in one) == [25, 25, 25, 25]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _comm(kind, rank, world, port, rdzv):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      BLUEICE_AMD_RDZV=rdzv)
    if kind == 'gloo':
        from comm_adapters import GlooCommunicator
        return GlooCommunicator(rank, world)
    from blueice_amd.comm import connect
    return connect(backend='socket')


def _worker(kind, rank, world, port, rdzv, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from blueice_amd.sharding import sharded_eval_points, sharded_eval_toys
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    comm = _comm(kind, rank, world, port, rdzv)
    try:
        m = SyntheticModel.named('mini3')
        dense = m.dense_model()
        counts = m.counts(dense=True)
        z, r = m.random_points(37, seed=2)
        calls = []

        def eval_fn(zz, rr):
            calls.append(len(zz))
            return orc.loglikelihood_batch(dense, counts, zz, rr)

        ll = sharded_eval_points(eval_fn, m.anchor_z, z, r, comm)
        toys = np.stack([m.counts(dense=True, dataset=t) for t in range(5)])
        z0, r0 = m.default_point()
        lt = sharded_eval_toys(lambda a, b: orc.loglikelihood_batch(dense, toys, np.tile(z0, (b - a, 1)),
                                                                     np.tile(r0, (b - a, 1)), dataset=np.arange(a, b)),
                               5, comm)
        mx = comm.all_reduce(np.array([float(rank), -float(rank)]), 'max')
        bits = comm.all_reduce(np.array([1 << rank, 0], dtype=np.int64), 'bor')
        q.put((rank, ll, lt, sum(calls), mx, bits, comm.broadcast_bytes(b'id-of-rank-0' if rank == 0 else None)))
    finally:
        comm.close()


def _run_ranks(target, kind, world, tmp_path, timeout=900):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    rdzv = str(tmp_path / ('rdzv_%s' % kind))
    procs = [ctx.Process(target=target, args=(kind, r, world, port, rdzv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize('kind', ['socket', 'gloo'])
def test_two_ranks_equal_single_process(kind, tmp_path):
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    got = _run_ranks(_worker, kind, 2, tmp_path)
    m = SyntheticModel.named('mini3')
    dense, counts = m.dense_model(), m.counts(dense=True)
    z, r = m.random_points(37, seed=2)
    want = orc.loglikelihood_batch(dense, counts, z, r)
    toys = np.stack([m.counts(dense=True, dataset=t) for t in range(5)])
    z0, r0 = m.default_point()
    want_t = orc.loglikelihood_batch(dense, toys, np.tile(z0, (5, 1)), np.tile(r0, (5, 1)), dataset=np.arange(5))
    assert sorted(g[0] for g in got) == [0, 1]
    for rank, ll, lt, n_eval, mx, bits, blob in got:
        np.testing.assert_array_equal(ll, want)            # every rank holds the full gathered vector
        np.testing.assert_array_equal(lt, want_t)
        assert 17 <= n_eval <= 20                          # each rank evaluated about half of the 37 points
        np.testing.assert_array_equal(mx, [1., 0.])
        np.testing.assert_array_equal(bits, [3, 0])
        assert blob == b'id-of-rank-0'


def _toy_points_worker(kind, rank, world, port, rdzv, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from blueice_amd.sharding import sharded_eval_toys_points
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    comm = _comm(kind, rank, world, port, rdzv)
    try:
        m = SyntheticModel.named('mini3')
        dense = m.dense_model()
        toys = np.stack([m.counts(dense=True, dataset=t) for t in range(6)])
        z, r = m.random_points(11, seed=9)
        z[4:8] = z[3]                                      # five hypotheses that differ in their rates only: one grid cell
        seen = []

        def eval_points(zz, rr):                           # the oracle in the role of bi_eval_datasets_points: -> [n, T]
            seen.append(zz.copy())
            return np.stack([orc.loglikelihood_batch(dense, toys, np.tile(zz[i], (6, 1)), np.tile(rr[i], (6, 1)), dataset=np.arange(6))
                             for i in range(len(zz))])

        out = sharded_eval_toys_points(eval_points, m.anchor_z, z, r, 6, comm)
        q.put((rank, out, np.concatenate(seen) if seen else np.zeros((0, m.d))))
    finally:
        comm.close()


@pytest.mark.parametrize('kind', ['socket', 'gloo'])
def test_toy_hypotheses_dealt_over_two_ranks(kind, tmp_path):
    """The multi-hypothesis toy-MC call sharded over ranks ("Next round" 1): every rank holds all datasets, the
    HYPOTHESES are dealt by grid cell, one gather of [n, T] blocks -- every rank ends with the full [P, T] matrix, each hypothesis
    was evaluated exactly once, and the hypotheses of one cell went to one rank."""
    from blueice_amd.sharding import cell_ids
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    got = sorted(_run_ranks(_toy_points_worker, kind, 2, tmp_path), key=lambda g: g[0])
    m = SyntheticModel.named('mini3')
    dense = m.dense_model()
    toys = np.stack([m.counts(dense=True, dataset=t) for t in range(6)])
    z, r = m.random_points(11, seed=9)
    z[4:8] = z[3]
    want = np.stack([orc.loglikelihood_batch(dense, toys, np.tile(z[i], (6, 1)), np.tile(r[i], (6, 1)), dataset=np.arange(6)) for i in range(11)])
    for rank, out, _ in got:
        np.testing.assert_array_equal(out, want)
    evaluated = np.concatenate([g[2] for g in got])
    assert len(evaluated) == 11 and 4 <= len(got[0][2]) <= 7          # each exactly once, shares balanced
    hot = cell_ids(m.anchor_z, z[3:4])[0]
    owners = [rank for rank, _, zs in got if len(zs) and (cell_ids(m.anchor_z, zs) == hot).any()]
    assert len(owners) == 1                                            # the rate scan of one cell stays on one rank


def test_three_socket_ranks(tmp_path):
    """An odd world size through the package's own communicator (unequal shares, star gather)."""
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    got = _run_ranks(_worker, 'socket', 3, tmp_path)
    m = SyntheticModel.named('mini3')
    z, r = m.random_points(37, seed=2)
    want = orc.loglikelihood_batch(m.dense_model(), m.counts(dense=True), z, r)
    assert sorted(g[0] for g in got) == [0, 1, 2]
    for g in got:
        np.testing.assert_array_equal(g[1], want)
        assert 11 <= g[3] <= 14


def test_eight_socket_ranks(tmp_path):
    """The driver's largest world size (N = 8) through the rendezvous, the sharded scan, the toy split (5 toys over 8
    ranks: three ranks get none) and the reductions."""
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    got = _run_ranks(_worker, 'socket', 8, tmp_path)
    m = SyntheticModel.named('mini3')
    z, r = m.random_points(37, seed=2)
    want = orc.loglikelihood_batch(m.dense_model(), m.counts(dense=True), z, r)
    assert sorted(g[0] for g in got) == list(range(8))
    assert sum(g[3] for g in got) == 37                       # every point evaluated exactly once
    for g in got:
        np.testing.assert_array_equal(g[1], want)
        assert len(g[2]) == 5
        np.testing.assert_array_equal(g[4], [7., 0.])
        np.testing.assert_array_equal(g[5], [255, 0])
        assert g[6] == b'id-of-rank-0'


def _bins_worker(kind, rank, world, port, rdzv, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from blueice_amd.sharding import bin_sharded_eval, split_range
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    comm = _comm(kind, rank, world, port, rdzv)
    try:
        m = SyntheticModel.named('mini3')
        dense, counts = m.dense_model(), m.counts(dense=True)
        lo, hi = split_range(m.B, rank, world)
        local = dict(anchor_z=dense['anchor_z'], ps=dense['ps'][..., lo:hi], mus=dense['mus'], n_model=None)

        class SliceCtx:                       # stands where a DeviceContext holding this rank's bins would
            bb_source = -1

            def eval(self, z, r):
                st = np.zeros(len(z), np.int32)
                st[rank] = 4 << rank          # a flag raised on this rank's slice only
                return orc.loglikelihood_batch(local, counts[lo:hi], z, r), st

        z, r = m.random_points(9, seed=3)
        ll, st = bin_sharded_eval(SliceCtx(), z, r, comm)
        q.put((rank, ll, st))
    finally:
        comm.close()


@pytest.mark.parametrize('kind', ['socket', 'gloo'])
def test_bin_sharding_allreduce_two_ranks(kind, tmp_path):
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    got = _run_ranks(_bins_worker, kind, 2, tmp_path)
    m = SyntheticModel.named('mini3')
    z, r = m.random_points(9, seed=3)
    want = orc.loglikelihood_batch(m.dense_model(), m.counts(dense=True), z, r)
    for rank, ll, st in got:
        np.testing.assert_allclose(ll, want, rtol=1e-12)
        np.testing.assert_array_equal(st[:3], [4, 8, 0])       # status bits are OR-ed over the ranks


def test_launcher_starts_ranks_and_propagates_failure(tmp_path):
    """python -m blueice_amd.launch: ranks find each other through the rendezvous file; a failing rank fails the run."""
    import subprocess
    script = tmp_path / 'ranks.py'
    script.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from blueice_amd.comm import connect\n"
        "c = connect(backend='socket')\n"
        "tot = c.all_reduce(np.array([float(c.rank + 1)]))\n"
        "assert tot[0] == c.world * (c.world + 1) / 2\n"
        "assert int(os.environ['LOCAL_RANK']) == c.rank\n"
        "c.close()\n"
        "sys.exit(3 if (len(sys.argv) > 1 and c.rank == 1) else 0)\n" % ROOT)
    env = dict(os.environ, PYTHONPATH=ROOT)
    ok = subprocess.run([sys.executable, '-m', 'blueice_amd.launch', '--nproc', '3', str(script)], env=env, timeout=300)
    assert ok.returncode == 0
    bad = subprocess.run([sys.executable, '-m', 'blueice_amd.launch', '--nproc', '2', str(script), 'fail'], env=env, timeout=300)
    assert bad.returncode == 3


# ---- the RCCL initialisation's failure modes, with a scripted stand-in for librccl (comm.py) ----------
class _FakeRccl:
    """What blueice_amd.comm calls on librccl.so, scripted per rank: `uid_rc` is ncclGetUniqueId's return code,
    `init` what ncclCommInitRank does ('ok', 'fail', 'hang')."""

    def __init__(self, uid_rc=0, init='ok'):
        self.uid_rc, self.init = uid_rc, init

    def ncclGetUniqueId(self, ref):
        return self.uid_rc

    def ncclGetErrorString(self, rc):
        return b'scripted failure %d' % rc

    def ncclCommInitRank(self, comm_ref, world, uid, rank):
        import time
        if self.init == 'hang':
            time.sleep(3600)
        return 0 if self.init == 'ok' else 5

    def ncclGetVersion(self, ref):
        return 0

    def ncclCommDestroy(self, comm):
        return 0


class _FakeCtx:
    stream = 0

    def device_alloc(self, n):
        class Buf:
            def free(self):
                pass
        return Buf()

    def sync(self):
        pass


def _rccl_failure_worker(kind, rank, world, port, rdzv, q):
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      BLUEICE_AMD_RDZV=rdzv)
    from blueice_amd import comm as cm
    script = {'uid_fails': _FakeRccl(uid_rc=3 if rank == 0 else 0),
              'init_fails_on_one': _FakeRccl(init='fail' if rank == 1 else 'ok'),
              'init_hangs_on_one': _FakeRccl(init='hang' if rank == 1 else 'ok'),
              'all_fine': _FakeRccl()}[kind]
    cm.load_rccl = lambda: script
    t = time.monotonic()
    try:
        c = cm.connect(_FakeCtx(), backend='rccl', timeout=3.0)
        # whatever came back must work as a communicator on every rank
        tot = c.boot.all_reduce(np.array([1.0])) if c.kind == 'rccl' else c.all_reduce(np.array([1.0]))
        q.put((rank, c.kind, getattr(c, 'fallback_reason', ''), float(tot[0]), time.monotonic() - t))
        c.boot.close() if c.kind == 'rccl' else c.close()
    except cm.CommInitTimeout as e:
        q.put((rank, 'timeout', str(e), float(e.stuck), time.monotonic() - t))
        q.close()
        q.join_thread()                              # the message is out before the process ends abruptly
        os._exit(0)                                  # (a rank with a stuck init thread leaves like this; bench.py: non-zero)


@pytest.mark.parametrize('kind', ['uid_fails', 'init_fails_on_one', 'init_hangs_on_one', 'all_fine'])
def test_rccl_init_failures_are_agreed_on(kind, tmp_path):
    got = sorted(_run_ranks(_rccl_failure_worker, kind, 2, tmp_path))
    assert [g[0] for g in got] == [0, 1]
    if kind == 'all_fine':
        assert all(g[1] == 'rccl' and g[3] == 2.0 for g in got)
    elif kind == 'uid_fails':
        # rank 0 could not draw an id: it says so, nobody waits for an id that never comes, both gather through sockets
        assert all(g[1] == 'socket' and 'ncclGetUniqueId failed' in g[2] and g[3] == 2.0 and g[4] < 2.5 for g in got), got
    elif kind == 'init_fails_on_one':
        assert all(g[1] == 'socket' and g[3] == 2.0 for g in got), got
        assert 'ncclCommInitRank failed' in got[1][2] and 'another rank' in got[0][2]
    else:
        # one rank never comes back from ncclCommInitRank: EVERY rank gives up together, only that one is stuck
        assert all(g[1] == 'timeout' for g in got), got
        assert [g[3] for g in got] == [0.0, 1.0]
        assert all(2.5 < g[4] < 12 for g in got), got


# ---- sharded_scan_device: a failure on ONE rank is agreed on by all -------------------------------------
class _ScriptedPlan:
    def __init__(self, values, fail_run):
        self.values, self.fail_run = values, fail_run

    def run(self, ptr=None):
        from blueice_amd.exceptions import DeviceError
        if self.fail_run:
            raise DeviceError('scripted HIP error on this rank')
        if ptr is not None:
            ptr.data[:len(self.values)] = self.values

    def status(self):
        return 0

    def read(self):
        return self.values, np.zeros(len(self.values), np.int32)

    def unsort(self, recv, stride, full):
        full.data[:] = np.concatenate([recv.data[r * stride:(r + 1) * stride] for r in range(len(recv.data) // stride)])[:len(full.data)]

    def close(self):
        pass


class _ScriptedBuf:
    def __init__(self, n):
        self.data = np.zeros(n // 8)
        self.ptr = self

    def from_host(self, a):
        a = np.ravel(a)
        self.data[:len(a)] = a

    def to_host(self, dtype, n):
        return self.data[:n].copy()


class _ScriptedCtx:
    """The calls sharded_scan_device makes on a DeviceContext; the 'evaluation' of point i is its index."""
    bb_source = -1

    def __init__(self, anchor_z, rank, mode):
        self.anchor_z, self.rank, self.mode = anchor_z, rank, mode

    def device_alloc(self, n):
        return _ScriptedBuf(n)

    def plan_share(self, z, r, dataset, rank, world):
        from blueice_amd.exceptions import DeviceError
        # what the library raises: a refusal is BI_ERR_INVALID -> ValueError (DeviceContext._check; tests/test_sharding_gpu.py
        # asserts exactly that for Beeston-Barlow batches that need exact totals), a failure of one rank is a DeviceError
        if self.mode == 'plan_refused_everywhere':
            raise ValueError('scripted: the device planner refuses this batch (needs the host planner)')
        if self.mode == 'plan_fails_on_one' and rank == 1:
            raise DeviceError('scripted: hipMalloc failed on this rank')
        n = len(z)
        lo, hi = rank * n // world, (rank + 1) * n // world
        return _ScriptedPlan(np.arange(lo, hi, dtype=float), self.mode == 'run_fails_on_one' and rank == 1)

    def plan(self, z, r, dataset=None):
        return _ScriptedPlan(z[:, 0].copy(), False)                        # host-dealt route: z carries the point's index


def _scan_failure_worker(kind, rank, world, port, rdzv, q):
    sys.path.insert(0, ROOT)
    from blueice_amd.exceptions import DeviceError
    from blueice_amd.sharding import sharded_scan_device
    comm = _comm('socket', rank, world, port, rdzv)
    try:
        grid = [np.array([0.0, 1000.0])]
        z = np.arange(40, dtype=float)[:, None]
        try:
            out, _ = sharded_scan_device(_ScriptedCtx(grid, rank, kind), z, np.ones((40, 1)), comm)
            q.put((rank, 'ok', out.tolist()))
        except DeviceError as e:
            q.put((rank, 'raised', str(e)))
        except BaseException as e:                        # (a bug in the test's stand-ins: report it instead of leaving the parent waiting)
            q.put((rank, 'crashed', repr(e)))
            raise
        comm.barrier()                                    # the communicator is still in step on every rank
    finally:
        comm.close()


@pytest.mark.parametrize('kind', ['all_fine', 'plan_refused_everywhere', 'plan_fails_on_one', 'run_fails_on_one'])
def test_scan_failures_on_one_rank_are_agreed_on(kind, tmp_path):
    got = sorted(_run_ranks(_scan_failure_worker, kind, 2, tmp_path, timeout=120))
    assert [g[0] for g in got] == [0, 1]
    if kind == 'run_fails_on_one':
        # nobody is left waiting in the gather: both ranks raise, the failing one its own error
        assert [g[1] for g in got] == ['raised', 'raised'], got
        assert 'another rank' in got[0][2] and 'scripted HIP error' in got[1][2]
    else:
        # no plan on some rank (refused for all, or failed on one): ALL ranks take the host-dealt route and answer
        assert all(g[1] == 'ok' and g[2] == list(map(float, range(40))) for g in got), got
