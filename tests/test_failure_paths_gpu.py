"""The error paths of the in-launch finish, taken on purpose ("What's weak" 6).

A work item's blocks post their partial sums into mailbox slots and the item's last block collects them
(blueice_amd/csrc/bi_dev_common.h: mail_post / mail_take).  The collector's wait is bounded; when it runs out the
result is nan with BI_ST_INTERNAL and the host empties the mailbox (reset_mail) before the next launch.  Two
injected faults (bi_set_param debug_skip_post / debug_late_post, consumed by the next mailbox launch):
  skip  block k never posts                  -> the collector times out; nothing is left behind
  late  block k posts after the collector    -> the collector times out AND a stale value stays in the mailbox:
        has given up                            only reset_mail makes the next call correct
Also: the parameter surface fails loudly on unknown names (no -1 that reads as a plausible number).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ST_INTERNAL = 32


@pytest.fixture()
def mini():
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_param('sparse', 0)                  # every bin visited: 9 tiles of 512 bins -> several blocks per item
    ctx.upload_counts(m.counts())
    ctx.set_param('mail_timeout_ms', 50)        # the test should not sit out the production 2 s
    yield m, ctx
    ctx.close()


def _oracle(m, z, r):
    from oracle import blueice_oracle as orc
    dense = m.dense_model()
    counts = m.counts()
    return np.array([orc.loglikelihood(dense, counts, z[i], r[i]) for i in range(len(z))])


@pytest.mark.parametrize('fault', ['debug_skip_post', 'debug_late_post'])
def test_single_call_times_out_and_recovers(mini, fault):
    m, ctx = mini
    z, r = m.random_points(3, seed=5)
    want = _oracle(m, z, r)
    good, st = ctx.eval_one(z[0], r[0])
    assert st == 0 and abs(good - want[0]) <= 1e-10 * abs(want[0])
    resets = ctx.get_param('n_mail_resets')
    ctx.set_param(fault, 0)                              # block 0 of the next launch misbehaves
    ll, st = ctx.eval_one(z[1], r[1])
    assert np.isnan(ll) and st & ST_INTERNAL, (ll, st)
    assert ctx.get_param('n_mail_resets') == resets + 1
    # the fault was consumed, the mailbox is empty again: the very next calls are right, bit for bit reproducible
    for i in (1, 2, 0):
        ll, st = ctx.eval_one(z[i], r[i])
        assert st == 0 and abs(ll - want[i]) <= 1e-10 * abs(want[i]), (i, ll, want[i])
    assert ctx.eval_one(z[0], r[0])[0] == good


@pytest.mark.parametrize('fault', ['debug_skip_post', 'debug_late_post'])
def test_batched_launch_times_out_and_recovers(mini, fault):
    """A batch in several grid cells (one work item each, in-launch finish): every item's block 1 misbehaves."""
    m, ctx = mini
    z, r = m.stratified_points(seed=2)                   # one point per cell
    z, r = z[:6], r[:6]
    want = _oracle(m, z, r)
    ctx.set_param('blocks_per_cu', 1)
    ctx.set_param(fault, 1)
    ll, st = ctx.eval(z, r)
    assert np.all(np.isnan(ll)) and np.all(st & ST_INTERNAL), (ll, st)
    ll, st = ctx.eval(z, r)
    assert not st.any()
    np.testing.assert_allclose(ll, want, rtol=1e-10)


def test_plan_status_reports_and_resets(mini):
    """Results left in HBM (bi_run_plan(out_dev) -> collective): nobody reads the status array, so the caller asks for
    its OR -- and a timed-out collector must not go unnoticed there (sharding.py:127)."""
    from blueice_amd.exceptions import DeviceError
    m, ctx = mini
    z, r = m.stratified_points(seed=4)
    z, r = z[:4], r[:4]
    want = _oracle(m, z, r)
    plan = ctx.plan(z, r)
    buf = ctx.device_alloc(8 * len(z))
    plan.run(buf.ptr)
    assert plan.status() == 0
    np.testing.assert_allclose(buf.to_host(np.float64, len(z)), want, rtol=1e-10)
    resets = ctx.get_param('n_mail_resets')
    ctx.set_param('debug_late_post', 0)
    plan.run(buf.ptr)
    with pytest.raises(DeviceError):
        plan.status()
    assert ctx.get_param('n_mail_resets') == resets + 1
    plan.run(buf.ptr)
    assert plan.status() == 0
    np.testing.assert_allclose(buf.to_host(np.float64, len(z)), want, rtol=1e-10)
    plan.close()


def test_parameter_surface_is_enumerable_and_fails_loudly(mini):
    m, ctx = mini
    params = ctx.list_params()
    assert params['sparse'] == 'rw' and params['n_scan_launches'] == 'r' and params['debug_skip_post'] == 'w'
    for name, access in params.items():
        if 'r' in access:
            ctx.get_param(name)                          # every listed readable name reads
    with pytest.raises(ValueError, match='unknown parameter'):
        ctx.get_param('n_scan_launchs')                  # a typo is an error, not -1
    with pytest.raises(ValueError, match='write-only'):
        ctx.get_param('debug_skip_post')
    with pytest.raises(ValueError, match='unknown parameter'):
        ctx.set_param('sprase', 1)
    with pytest.raises(ValueError, match='read-only'):
        ctx.set_param('n_scan_launches', 0)
    with pytest.raises(ValueError):
        ctx.set_param('sparse', 7)                       # range check


def test_device_buffers_go_with_the_context():
    from blueice_amd.device import DeviceContext
    ctx = DeviceContext(0)
    a = ctx.device_alloc(1 << 20)
    b = ctx.device_alloc(1 << 20)
    assert ctx.get_param('user_allocations') == 2
    a.free()
    assert ctx.get_param('user_allocations') == 1
    with pytest.raises(ValueError):                      # not one of ours
        ctx._check(ctx._lib.bi_device_free(ctx._h, 12345 * 4096))
    ctx.close()                                          # frees b
    b.free()                                             # a no-op on a closed context


def test_recycle_cache_is_visible_and_can_be_dropped():
    """The context parks freed plan / scratch buffers for reuse (hipMalloc + hipFree of a 10^6-point plan's buffers cost
    2 ms per step): the parked bytes are readable, bounded, given back on request, and results do not depend on it."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.upload_counts(m.counts(dense=True))
    z, r = m.random_points(20000, seed=2)
    first = ctx.eval(z, r)[0]
    parked = ctx.get_param('recycle_cache_bytes')
    assert 0 < parked <= 4 << 30
    again = ctx.eval(z, r)[0]                                 # runs on recycled buffers
    np.testing.assert_array_equal(again, first)
    ctx.set_param('drop_recycle_cache', 1)
    assert ctx.get_param('recycle_cache_bytes') == 0
    np.testing.assert_array_equal(ctx.eval(z, r)[0], first)
    assert ctx.list_params()['drop_recycle_cache'] == 'w' and ctx.list_params()['recycle_cache_bytes'] == 'r'
    ctx.close()
