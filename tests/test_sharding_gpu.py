"""Ranks on the GPU box (the box has one GPU, so every rank uses device 0): the sharded scan / toy-MC run through
the real HIP path equals the single-process device result and the oracle; RCCL bound directly (one-rank
communicator: the real ncclCommInitRank / ncclAllGather / ncclAllReduce on the context's stream); and two ranks
asking for RCCL on ONE GPU, which RCCL refuses -- the agreed fall-back to the socket gather must then carry the run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
backend, out = sys.argv[2], sys.argv[3]
from blueice_amd.comm import connect
from blueice_amd.device import DeviceContext
from blueice_amd.sharding import sharded_eval_points, sharded_eval_toys, sharded_scan_device, split_range
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('mini3')
ctx = DeviceContext(0)
comm = connect(ctx, backend=backend)
m.upload(ctx)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(301, seed=4)
ll = sharded_eval_points(lambda zz, rr: ctx.eval(zz, rr)[0], m.anchor_z, z, r, comm)
ll_dev, rerun = sharded_scan_device(ctx, z, r, comm)
ll_dev2 = rerun()
# toy-MC: every rank draws ITS range of one ensemble of 9 toys (toy_offset) and evaluates it
z0, r0 = m.default_point()
t0, t1 = split_range(9, comm.rank, comm.world)
ctx.set_param('toy_offset', t0)
ctx.generate_toys(z0, r0, t1 - t0, seed=77)
lt = sharded_eval_toys(lambda a, b: ctx.eval_datasets(z0, r0, 0, b - a)[0], 9, comm)
first = ctx.download_counts(0)
tot = comm.all_reduce(np.array([float(comm.rank + 1)]))
bits = comm.all_reduce(np.array([1 << comm.rank], dtype=np.int64), 'bor')
np.savez(out + '.%d.npz' % comm.rank, ll=ll, ll_dev=ll_dev, ll_dev2=ll_dev2, lt=lt, first=first, tot=tot, bits=bits,
         kind=comm.kind, reason=getattr(comm, 'fallback_reason', ''), t0=t0)
comm.close()
ctx.close()
"""


def _launch(tmp_path, nproc, backend):
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_SCRIPT)
    out = str(tmp_path / ('res_%s_%d' % (backend, nproc)))
    res = subprocess.run([sys.executable, '-m', 'blueice_amd.launch', '--nproc', str(nproc), '--devices',
                          ','.join(['0'] * nproc), str(script), ROOT, backend, out],
                         env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return [np.load(out + '.%d.npz' % r) for r in range(nproc)]


def _single_process_truth():
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    counts = m.counts(dense=True)
    ctx.upload_counts(counts)
    z, r = m.random_points(301, seed=4)
    single = ctx.eval(z, r)[0]
    want = orc.loglikelihood_batch(m.dense_model(), counts, z[:12], r[:12])
    z0, r0 = m.default_point()
    ctx.generate_toys(z0, r0, 9, seed=77)
    toys = np.stack([ctx.download_counts(t) for t in range(9)])
    single_t = ctx.eval_datasets(z0, r0)[0]
    want_t = orc.loglikelihood(m.dense_model(), toys[4], z0, r0)
    ctx.close()
    return single, want, toys, single_t, want_t


def _check(got, world):
    single, want, toys, single_t, want_t = _single_process_truth()
    assert abs(single_t[4] - want_t) <= 1e-10 * abs(want_t)
    for g in got:
        for key in ('ll', 'll_dev', 'll_dev2'):
            np.testing.assert_allclose(g[key], single, rtol=1e-13)
        np.testing.assert_allclose(g['ll'][:12], want, rtol=1e-10)
        np.testing.assert_allclose(g['lt'], single_t, rtol=1e-13)         # the ranks' toys ARE the one-process toys
        np.testing.assert_array_equal(g['first'], toys[int(g['t0'])])
        assert g['tot'][0] == world * (world + 1) / 2 and g['bits'][0] == (1 << world) - 1


def test_two_socket_ranks_on_gpu_equal_single_process(tmp_path):
    got = _launch(tmp_path, 2, 'socket')
    assert all(str(g['kind']) == 'socket' for g in got)
    _check(got, 2)


def test_rccl_bound_directly_one_rank(tmp_path):
    """The real thing with the one GPU the box has: ncclCommInitRank(nranks = 1), the gather between device buffers
    on the context's stream (`sharded_scan_device` is given a communicator with all_gather_device, but takes its
    device route only for world > 1, so the collectives are also called explicitly)."""
    script = tmp_path / 'one.py'
    script.write_text(r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from blueice_amd.comm import connect
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('mini3')
ctx = DeviceContext(0)
comm = connect(ctx, backend='rccl', rank=0, world=1)
assert comm.kind == 'rccl', getattr(comm, 'fallback_reason', '')
print('RCCL version', comm.version)
# the communicator's own account of itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): what a run reports
assert (comm.n_ranks, comm.user_rank, comm.device) == (1, 0, 0), (comm.n_ranks, comm.user_rank, comm.device)
from blueice_amd.comm import describe
d = describe(comm, 'rccl')
assert d['rccl_ranks'] == 1 and d['rank_devices'] == [0] and d['rank_devices_source'] == 'ncclCommCuDevice' and d['gather_fallback_reason'] is None, d
m.upload(ctx)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(100, seed=8)
want, _ = ctx.eval(z, r)
plan = ctx.plan(z, r)
send, recv = ctx.device_alloc(8 * 100), ctx.device_alloc(8 * 100)
recv.from_host(np.full(100, np.nan))
plan.run(send.ptr)                                   # bi_run_plan writes into the gather's send buffer ...
comm.all_gather_device(send.ptr, recv.ptr, 100)      # ... and RCCL moves it on the same stream
np.testing.assert_array_equal(recv.to_host(), want)
comm.all_reduce_device(send.ptr, recv.ptr, 100, 'sum')
np.testing.assert_array_equal(recv.to_host(), want)
np.testing.assert_array_equal(comm.all_gather(np.arange(5.0)), np.arange(5.0)[None])
np.testing.assert_array_equal(comm.all_reduce(np.array([3, 5], dtype=np.int64), 'max'), [3, 5])
np.testing.assert_array_equal(comm.all_reduce(np.array([6], dtype=np.int32), 'bor'), [6])
comm.barrier()
comm.close()
ctx.close()
print('RCCL_ONE_RANK_OK')
""")
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PYTHONPATH=ROOT))
    assert res.returncode == 0 and 'RCCL_ONE_RANK_OK' in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]


def test_two_ranks_asking_for_rccl_on_one_gpu(tmp_path):
    """RCCL refuses two ranks on one device; `connect` must notice on every rank, agree, and hand back the socket
    communicator -- or, should a future RCCL accept it, the device route must give the same numbers."""
    got = _launch(tmp_path, 2, 'rccl')
    kinds = {str(g['kind']) for g in got}
    assert len(kinds) == 1 and kinds <= {'rccl', 'socket'}
    if kinds == {'socket'}:
        assert all(str(g['reason']) for g in got)
    _check(got, 2)


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


@pytest.mark.parametrize('sparse', [0, 1])
def test_scan_dealt_on_the_device_equals_one_plan(sparse):
    """bi_plan_points_share / bi_plan_unsort in one process: the `world` shares of a scan, evaluated one after the other
    on one context and 'gathered' by hand, reproduce the plain batched evaluation -- rejected points
    (outside the anchor box, nan, negative rates) included; shares are contiguous, balanced and cover every valid point."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(m.counts(dense=not sparse))
    z, r = m.random_points(2999, seed=12)
    z[5, 1] = 7.0            # outside the box
    z[77, 0] = np.nan
    r[300, 2] = -1.0         # unphysical
    want, want_st = ctx.eval(z, r)
    assert np.isneginf(want[[5, 77, 300]]).all() and np.isfinite(np.delete(want, [5, 77, 300])).all()
    P = len(z)
    plain = ctx.plan(z, r)
    with pytest.raises(ValueError):                         # a plain plan is not a share
        ctx._check(ctx._lib.bi_plan_share_info(plain._h, None, None, None))
    plain.close()
    for world in (2, 3, 5, 8):
        stride = -(-P // world)
        send = ctx.device_alloc(8 * stride)
        recv = ctx.device_alloc(8 * stride * world)
        full = ctx.device_alloc(8 * P)
        recv.from_host(np.full(stride * world, np.nan))
        spans, plans = [], []
        for rank in range(world):
            plan = ctx.plan_share(z, r, None, rank, world)
            assert plan.n_valid == P - 3
            spans.append(plan.share)
            plan.run(send.ptr)
            assert plan.status() & ~3 == 0                  # only the rejected points' bits
            n = plan.share[1] - plan.share[0]
            recv.from_host(send.to_host(np.float64, n), offset_bytes=8 * stride * rank)
            plans.append(plan)
        assert spans[0][0] == 0 and spans[-1][1] == P - 3 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
        plans[-1].unsort(recv.ptr, stride, full.ptr)        # any rank's plan holds the same sorted -> original map
        got = full.to_host(np.float64, P)
        # (a share chops a cell's points into other work items than the whole list does: equal to rounding)
        assert np.isneginf(got[[5, 77, 300]]).all()
        keep = np.isfinite(want)
        np.testing.assert_allclose(got[keep], want[keep], rtol=1e-12)
        with pytest.raises(ValueError):                     # sorted-order results are not readable as a point vector
            plans[0].read()
        for p in plans:
            p.close()
        for b in (send, recv, full):
            b.free()
    # fewer valid points than ranks: some shares are empty, and a scan whose every point is rejected has no share at all
    for zz, rr, n_ok in ((z[:4], r[:4], 4), (z[5:6], r[5:6], 0)):
        world = 8
        stride = 1
        send, recv, full = ctx.device_alloc(8), ctx.device_alloc(8 * world), ctx.device_alloc(8 * len(zz))
        recv.from_host(np.full(world, np.nan))
        plans = []
        for rank in range(world):
            plan = ctx.plan_share(zz, rr, None, rank, world)
            assert plan.n_valid == n_ok and plan.share[1] - plan.share[0] == (1 if rank < n_ok else 0)
            plan.run(send.ptr)
            plan.status()
            if plan.share[1] > plan.share[0]:
                recv.from_host(send.to_host(np.float64, 1), offset_bytes=8 * rank)
            plans.append(plan)
        plans[0].unsort(recv.ptr, stride, full.ptr)
        got = full.to_host(np.float64, len(zz))
        ref = ctx.eval(zz, rr)[0]
        np.testing.assert_allclose(got[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=1e-12)
        assert np.array_equal(np.isneginf(got), np.isneginf(ref))
        for p in plans:
            p.close()
        for b in (send, recv, full):
            b.free()
    ctx.close()


@pytest.mark.parametrize('sparse', [0, 1])
def test_points_resident_in_hbm_equal_host_points(sparse):
    """bi_plan_points_resident: the points are read where they lie in HBM.  Whole batch: the very bits of the plan made from
    host arrays (same planner, same work items); a share: the bits of bi_plan_points_share; with datasets; without rate
    scales; what the device planner cannot do is an error, and so is a pointer that is not device memory."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_param('sparse', sparse)
    T = 3
    rng = np.random.default_rng(8)
    counts = np.stack([m.counts(dense=not sparse) for _ in range(T)])
    counts[1] = rng.permutation(counts[1])
    counts[2, ::7] += 1
    ctx.upload_counts(counts)
    P = 2500
    z, r = m.random_points(P, seed=31)
    z[9, 0] = -9.0
    r[40, 1] = np.nan
    ds = rng.integers(0, T, P).astype(np.int64)
    ds[100] = 17                                            # no such dataset
    dz, dr, dd = ctx.device_alloc(z.nbytes), ctx.device_alloc(r.nbytes), ctx.device_alloc(ds.nbytes)
    dz.from_host(z); dr.from_host(r); dd.from_host(ds)
    for use_r, use_ds in ((True, True), (True, False), (False, False)):
        a = ctx.plan(z, r if use_r else None, ds if use_ds else None)
        b = ctx.plan_resident(P, dz, dr if use_r else None, dd if use_ds else None)
        assert b.bytes == a.bytes and b.launches == a.launches
        (la, sa), (lb, sb) = a.run().read(), b.run().read()
        np.testing.assert_array_equal(lb, la)
        np.testing.assert_array_equal(sb, sa)
        assert np.isneginf(lb[9]) and (not use_r or np.isneginf(lb[40])) and (not use_ds or np.isneginf(lb[100]))
        assert np.isfinite(np.delete(lb, [9, 40, 100])).all()
        a.close(); b.close()
    # a share of a dealt scan
    world = 3
    stride = -(-P // world)
    out_a, out_b = ctx.device_alloc(8 * stride), ctx.device_alloc(8 * stride)
    for rank in range(world):
        a = ctx.plan_share(z, r, ds, rank, world)
        b = ctx.plan_resident(P, dz, dr, dd, rank, world)
        assert (b.n_valid, b.share) == (a.n_valid, a.share)
        a.run(out_a.ptr); b.run(out_b.ptr)
        assert a.status() == b.status()
        n = a.share[1] - a.share[0]
        np.testing.assert_array_equal(out_b.to_host(np.float64, n), out_a.to_host(np.float64, n))
        a.close(); b.close()
    # the buffers may go as soon as the plan exists
    b = ctx.plan_resident(P, dz, dr, dd)
    want = ctx.eval(z, r, ds)[0]
    dz.from_host(np.zeros_like(z)); dr.free()
    np.testing.assert_array_equal(b.run().read()[0], want)
    b.close()
    dr = ctx.device_alloc(r.nbytes)
    dr.from_host(r); dz.from_host(z)
    # errors: host memory, too small a buffer, an empty batch is fine
    with pytest.raises(ValueError, match='not device memory'):
        ctx.plan_resident(P, z.ctypes.data, dr, dd)
    with pytest.raises(ValueError, match='not device memory'):
        ctx.plan_resident(P, dz, dr, ds.ctypes.data)
    with pytest.raises(ValueError, match='holds'):
        ctx.plan_resident(P + 1, dz, dr, dd)
    with pytest.raises(ValueError, match='share'):
        ctx.plan_resident(P, dz, dr, dd, 3, 3)
    e = ctx.plan_resident(0, dz)
    assert len(e.run().read()[0]) == 0
    e.close()
    ctx.close()
    # the extended unbinned likelihood (rows = pdf values at the events): the same planner, the same bits
    cu = DeviceContext(0)
    m.upload(cu)
    cu.set_unbinned(1e-12)
    zu, ru = m.random_points(1500, seed=3)
    uz, ur = cu.device_alloc(zu.nbytes), cu.device_alloc(ru.nbytes)
    uz.from_host(zu); ur.from_host(ru)
    a, b = cu.plan(zu, ru), cu.plan_resident(1500, uz, ur)
    np.testing.assert_array_equal(b.run().read()[0], a.run().read()[0])
    assert np.isfinite(b.read()[0]).all()
    a.close(); b.close()
    cu.close()
    # (Beeston-Barlow models and sources that may go negative are planned on the device as well since round 4: the tests
    #  below; what is still refused -- with a reason -- are batches that need the host: exact Beeston-Barlow totals,
    #  infinite rates)


# ---- the device planner takes Beeston-Barlow and negative-rate models ("Next round" 7) -------------------
def _bb_context(zero_u=False):
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini4bb', bb_source=0)
    ctx = DeviceContext(0)
    if zero_u:
        # exact zeros in the other sources' templates: bins with U_b == 0 become possible, the reference's first-root
        # assertion then hangs on the last bits of N(z)
        dense = m.dense_model()
        ps = dense['ps'].copy()
        ps[..., 1:, 3] = 0.0
        ps[..., 1:, 7] = 0.0
        ctx.upload_model(dense['anchor_z'], ps, dense['mus'], dense['n_model'], bb_source=0)
    else:
        m.upload(ctx)
    ctx.set_param('sparse', 0)
    ctx.upload_counts(m.counts(dense=True))
    return m, ctx


def test_beeston_barlow_scan_dealt_on_the_device_equals_one_plan():
    """bi_plan_points_share / bi_plan_points_resident on a Beeston-Barlow model (blueice/likelihood.py:618-660): work items of
    bb_max_group points, N(z) and p_cal per point from the per-anchor totals -- against the host planner (small batches) and
    the single-point kernel, assertion status bits included."""
    m, ctx = _bb_context()
    z, r = m.random_points(1500, seed=5)
    z[11, 0] = 9.0                         # outside the box
    r[12, 1] = -2.0                        # unphysical
    P = len(z)
    ctx.set_param('device_plan_min', 0)    # the host planner (8 points per work item, exact totals where needed)
    want, want_st = ctx.eval(z, r)
    ctx.set_param('device_plan_min', 512)
    got, got_st = ctx.eval(z, r)           # the same batch, now planned on the device
    np.testing.assert_array_equal(got_st, want_st)
    keep = np.isfinite(want)
    assert keep.sum() == P - 2 and np.isneginf(got[[11, 12]]).all()
    np.testing.assert_allclose(got[keep], want[keep], rtol=1e-12)
    for i in (0, 500, 1499):
        one, st1 = ctx.eval(z[i], r[i])
        assert st1[0] == got_st[i] and abs(one[0] - got[i]) <= 1e-12 * abs(one[0])
    for world in (2, 3):
        stride = -(-P // world)
        send, recv, full = ctx.device_alloc(8 * stride), ctx.device_alloc(8 * stride * world), ctx.device_alloc(8 * P)
        recv.from_host(np.full(stride * world, np.nan))
        spans = []
        for rank in range(world):
            plan = ctx.plan_share(z, r, None, rank, world)
            assert plan.n_valid == P - 2
            spans.append(plan.share)
            plan.run(send.ptr)
            assert plan.status() & ~(3 | 4 | 8) == 0
            recv.from_host(send.to_host(np.float64, plan.share[1] - plan.share[0]), offset_bytes=8 * stride * rank)
            last = plan
        assert spans[0][0] == 0 and spans[-1][1] == P - 2 and max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
        last.unsort(recv.ptr, stride, full.ptr)
        shared = full.to_host(np.float64, P)
        np.testing.assert_allclose(shared[keep], want[keep], rtol=1e-12)
        assert np.isneginf(shared[[11, 12]]).all()
    # points already in HBM
    bz, br = ctx.device_alloc(z.nbytes), ctx.device_alloc(r.nbytes)
    bz.from_host(z); br.from_host(r)
    plan = ctx.plan_resident(P, bz, br)
    plan.run()
    res, res_st = plan.read()
    np.testing.assert_allclose(res[keep], want[keep], rtol=1e-12)
    np.testing.assert_array_equal(res_st, want_st)
    ctx.close()


def test_device_planner_leaves_exact_totals_to_the_host():
    """Where some bin can have U_b == 0 the device planner steps back: large host-array batches silently take the host
    planner (same values, same assertion bits as small ones), shares and resident points are refused with a reason -- and
    sharded_scan_device then deals on the host (tests/test_sharding.py scripts that agreement)."""
    m, ctx = _bb_context(zero_u=True)
    z, r = m.random_points(1200, seed=6)
    ctx.set_param('device_plan_min', 0)
    want, want_st = ctx.eval(z, r)
    ctx.set_param('device_plan_min', 512)
    got, got_st = ctx.eval(z, r)
    np.testing.assert_array_equal(got_st, want_st)
    ok = np.isfinite(want)
    np.testing.assert_array_equal(got[ok], want[ok])              # the same planner in the end: the same bits
    with pytest.raises(ValueError, match='exact totals'):
        ctx.plan_share(z, r, None, 0, 2)
    bz, br = ctx.device_alloc(z.nbytes), ctx.device_alloc(r.nbytes)
    bz.from_host(z); br.from_host(r)
    with pytest.raises(ValueError, match='exact totals'):
        ctx.plan_resident(len(z), bz, br)
    ctx.close()


_BB_RANK_SCRIPT = """
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from blueice_amd.comm import connect
from blueice_amd.device import DeviceContext
from blueice_amd.sharding import sharded_scan_device
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('mini4bb', bb_source=0)
ctx = DeviceContext(0)
comm = connect(ctx, backend='socket')
dense = m.dense_model()
ps = dense['ps'].copy()
ps[..., 1:, 3] = 0.0
ps[..., 1:, 7] = 0.0
ctx.upload_model(dense['anchor_z'], ps, dense['mus'], dense['n_model'], bb_source=0)
ctx.set_param('sparse', 0)
ctx.upload_counts(m.counts(dense=True))
z, r = m.random_points(1200, seed=6)
want, want_st = ctx.eval(z, r)
got, rerun = sharded_scan_device(ctx, z, r, comm)
np.savez(sys.argv[2] + '.%d.npz' % comm.rank, want=want, got=got, again=rerun())
comm.close()
ctx.close()
"""


def test_sharded_scan_of_a_batch_the_device_planner_refuses(tmp_path):
    """bi_plan_points_share answers a Beeston-Barlow batch that needs exact totals with BI_ERR_INVALID, which
    DeviceContext raises as ValueError -- sharded_scan_device must take the agreed host-dealt route on every rank (and return the
    values of ctx.eval), not let the refusal escape on every rank."""
    script = tmp_path / 'bb_rank.py'
    script.write_text(_BB_RANK_SCRIPT)
    out = str(tmp_path / 'bb_res')
    res = subprocess.run([sys.executable, '-m', 'blueice_amd.launch', '--nproc', '2', '--devices', '0,0', str(script), ROOT, out],
                         env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for rank in range(2):
        g = np.load(out + '.%d.npz' % rank)
        ok = np.isfinite(g['want'])
        assert ok.sum() > 1000
        np.testing.assert_array_equal(g['got'][ok], g['want'][ok])          # the host planner either way: the same bits
        np.testing.assert_array_equal(g['again'][ok], g['want'][ok])
        np.testing.assert_array_equal(np.isnan(g['got']), np.isnan(g['want']))


def test_device_planner_takes_sources_that_may_go_negative():
    """allow_negative sources (blueice/likelihood.py:403-415): finite rates of either sign are planned on the device, shares and
    resident points included; an INFINITE rate is the host's to answer, and the planner says so."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_allow_negative([0, 1, 0, 0])
    ctx.set_param('sparse', 0)
    ctx.upload_counts(m.counts(dense=True))
    z, r = m.random_points(1400, seed=8)
    r[::7, 1] = -0.05                       # a little negative: legal for source 1
    ctx.set_param('device_plan_min', 0)
    want, want_st = ctx.eval(z, r)
    ctx.set_param('device_plan_min', 512)
    bz, br = ctx.device_alloc(z.nbytes), ctx.device_alloc(r.nbytes)
    bz.from_host(z); br.from_host(r)
    plan = ctx.plan_resident(len(z), bz, br)
    plan.run()
    got, got_st = plan.read()
    np.testing.assert_array_equal(got_st, want_st)
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-12)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    r2 = r.copy()
    r2[3, 1] = np.inf
    br.from_host(r2)
    with pytest.raises(ValueError, match='infinite rate'):
        ctx.plan_resident(len(z), bz, br)
    with pytest.raises(ValueError, match='infinite rate'):
        ctx.plan_share(z, r2, None, 0, 2)
    both, _ = ctx.eval(z, r2)               # host arrays: answered on the host the reference's way
    assert np.array_equal(np.isfinite(both[ok & (np.arange(len(z)) != 3)]), np.ones((ok & (np.arange(len(z)) != 3)).sum(), bool))
    ctx.close()
