"""Full-size parity against numbers produced by the REAL reference (tests/golden/fullsize.json, made by
tests/golden/make_golden_fullsize.py in the development container: blueice's own BinnedLogLikelihood prepared at
BASELINE.json's sizes through a plug-in Source, blueice/likelihood.py:318-427 + blueice/source.py:266-267), and the
BASELINE.json configurations at their STATED sizes: 10^4 toys (configs[2]), 10^6 scan points (configs[3]), the
625-anchor Beeston-Barlow model with 50^4 bins (configs[4]).  Tolerance: |device - reference| <= 1e-10 max(1, |ref|)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10
HERE = os.path.dirname(os.path.abspath(__file__))


def _golden(name):
    with open(os.path.join(HERE, 'golden', 'fullsize.json')) as f:
        return json.load(f)['cases'][name]


def _check_inputs(m, case):
    """The synthetic tensors regenerated here must be the ones the reference saw (same numpy random streams)."""
    ck = case['checks']
    a0 = m.anchor_ps(0)
    assert [float(v) for v in a0[0, :3]] == ck['ps_anchor0_first'] and float(a0.sum()) == ck['ps_anchor0_sum']
    assert [float(v) for v in m.anchor_mus(0)] == ck['mus_anchor0']
    if m.bb_source >= 0:
        assert float(m.anchor_n_model(0).sum()) == ck['n_model_anchor0_sum']


def _compare(ctx, m, case, modes):
    for dense in (False, True):
        key = 'dense' if dense else 'sparse'
        counts = m.counts(dense=dense)
        assert float(counts.sum()) == case['checks']['counts_%s_sum' % key]
        assert int(np.count_nonzero(counts)) == case['checks']['counts_%s_nonzero' % key]
        calls = [c for c in case['calls'] if c['data'] == key]
        assert len(calls) >= 3
        z = np.array([c['z'] for c in calls])
        r = np.array([c['mult'] for c in calls])
        for sparse in modes:
            ctx.set_param('sparse', sparse)
            ctx.upload_counts(counts)
            batch, bst = ctx.eval(z, r)
            for j, c in enumerate(calls):
                assert not c['reference_asserted']
                one, st = ctx.eval(z[j], r[j])
                for got in (one[0], batch[j]):
                    assert abs(got - c['ll']) <= RTOL * max(1.0, abs(c['ll'])), (key, sparse, c['label'], got, c['ll'])
                assert st[0] == 0 and bst[j] == 0


@pytest.fixture(scope='module')
def c2():
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('C2')
    ctx = DeviceContext(0)
    m.upload(ctx, threads=8)
    yield m, ctx
    ctx.close()


def test_c2_equals_the_reference_at_full_size(c2):
    """configs[1]: every kernel path (single-point, batched, non-empty-bin form) against the reference's lf(**kw)
    at an off-grid point, an anchor corner, the top edge, a switched-off source and an interior anchor plane."""
    m, ctx = c2
    case = _golden('C2')
    _check_inputs(m, case)
    _compare(ctx, m, case, modes=(0, 1))
    ctx.set_param('sparse', 1)


def test_c5_cell_beeston_barlow_equals_the_reference():
    """One grid cell of configs[4] (2^4 anchors, 6 sources, 50^4 bins, Beeston-Barlow on source 0: all 113 streams of
    an evaluation) against the reference's adjust_expectations + _compute_likelihood (blueice/likelihood.py:618-675)."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('C5-2anchor', bb_source=0)
    case = _golden('C5-2anchor')
    _check_inputs(m, case)
    ctx = DeviceContext(0)
    m.upload(ctx, threads=8)
    _compare(ctx, m, case, modes=(0,))
    # the same reference scalars through the matrix-core Beeston-Barlow scan kernel (k_scan_bb, csrc/bi_k_scan_bb.h: batches of at
    # least 64 points planned on the device; variant <20, 4>: 80 streams into U, 16 corners): each golden point rides in a batch of
    # 96 points of the cell -- and must come out as the reference has it, its status 0
    for dense in (False, True):
        key = 'dense' if dense else 'sparse'
        calls = [c for c in case['calls'] if c['data'] == key]
        ctx.upload_counts(m.counts(dense=dense))
        z, r = m.random_points(96, seed=77)
        for j, c in enumerate(calls):
            z[7 * j + 3], r[7 * j + 3] = c['z'], c['mult']
        ctx.set_param('device_plan_min', 1)
        before = ctx.get_param('n_bb_scan_launches')
        got, st = ctx.eval(z, r)
        assert ctx.get_param('n_bb_scan_launches') == before + 1
        ctx.set_param('scan_bb', 0)
        vec, st_v = ctx.eval(z, r)                       # the vector kernel (k_morph_reduce<8, true>) on the same batch
        ctx.set_param('scan_bb', 1)
        ctx.set_param('device_plan_min', 512)
        assert ctx.get_param('n_bb_scan_launches') == before + 1
        np.testing.assert_array_equal(st, st_v)
        np.testing.assert_allclose(got, vec, rtol=1e-12, atol=0)
        for j, c in enumerate(calls):
            assert st[7 * j + 3] == 0 and abs(got[7 * j + 3] - c['ll']) <= RTOL * abs(c['ll']), (key, c['label'], got[7 * j + 3], c['ll'])
    ctx.close()


def test_c3_ten_thousand_toys(c2):
    """configs[2] at its stated size: 10^4 toy datasets drawn on the device and evaluated by one call; 8 of them are
    fetched back and checked against the oracle, and the per-dataset point form agrees on a sample."""
    from oracle import blueice_oracle as orc
    m, ctx = c2
    T = 10000
    z, r = m.default_point()
    ctx.set_param('sparse', 1)
    ctx.generate_toys(z, r, T, seed=2024)
    z_eval = np.clip(z + 0.04, -2, 2)
    ll, st = ctx.eval_datasets(z_eval, r)
    assert st == 0 and ll.shape == (T,) and np.all(np.isfinite(ll))
    cell = m.cell_model(z_eval)
    picks = [0, 1, 17, 2500, 4999, 5000, 7777, T - 1]
    for t in picks:
        counts = ctx.download_counts(t)
        assert 8000 < counts.sum() < 12000 and np.all(counts == np.floor(counts))
        want = orc.loglikelihood(cell, counts, z_eval, r)
        assert abs(ll[t] - want) <= RTOL * abs(want), (t, ll[t], want)
    buf = ctx.device_alloc(8 * T)
    assert ctx.eval_datasets_device(buf.ptr, z_eval, r) == 0
    np.testing.assert_array_equal(buf.to_host(), ll)           # results left in HBM for the gather: same numbers
    buf.free()
    # the call above ran the tiled kernel (bin tiles of log mu staged through LDS over the tile-major lists); the row
    # kernel (one block per dataset gathering from the 8 MB table) must give the same sums in another order, and ranges
    # of datasets must be the same numbers as the whole
    ctx.set_param('dot_tiled', 0)
    rows, _ = ctx.eval_datasets(z_eval, r)
    ctx.set_param('dot_tiled', 1)
    np.testing.assert_allclose(rows, ll, rtol=1e-13, atol=0)
    part, _ = ctx.eval_datasets(z_eval, r, 2500, 7800)
    np.testing.assert_array_equal(part, ll[2500:7800])
    # (the per-dataset POINT form needs compacted templates: 41 MB per toy, beyond the budget at 10^4 toys -- it is
    # covered at T = 48 in test_fullsize_gpu.py::test_c3_toy_batch; here the library must say so, not guess)
    from blueice_amd.exceptions import NotPreparedException
    with pytest.raises(NotPreparedException):
        ctx.eval(z_eval, r, dataset=[17])


def test_c3_ten_thousand_toys_at_several_hypotheses(c2):
    """configs[2] over several parameter points in ONE call (bi_eval_datasets_points; the reference's double loop of
    blueice/inference.py:392-443 around blueice/model.py:69-91): 10^4 device-drawn toys x 6 hypotheses -- four rate hypotheses
    in one grid cell (one shared pass over its templates), two more in other cells -- against the oracle on fetched toys and
    against the single-point call."""
    from oracle import blueice_oracle as orc
    m, ctx = c2
    T = 10000
    z, r = m.default_point()
    ctx.set_param('sparse', 1)
    ctx.generate_toys(z, r, T, seed=2025)
    P = 6
    zs = np.repeat(z[None, :], P, axis=0)
    rs = np.repeat(r[None, :], P, axis=0)
    rs[:4, 0] *= (0.5, 1.0, 1.5, 2.0)                       # a signal-strength scan: same cell
    zs[4] = np.clip(z + 0.9, -2, 2)                          # two more hypotheses in other cells
    zs[5] = np.clip(z - 1.1, -2, 2)
    before = ctx.get_param('n_toy_points_passes')
    ll, st = ctx.eval_datasets_points(zs, rs)
    assert ctx.get_param('n_toy_points_passes') == before + 2 and ctx.get_param('tmm_entry_bytes') == 2
    assert ll.shape == (P, T) and not st.any() and np.all(np.isfinite(ll))
    picks = [0, 1, 2500, 4999, 7777, T - 1]
    fetched = {t: ctx.download_counts(t) for t in picks}
    for p in range(P):
        cell = m.cell_model(zs[p])
        for t in picks:
            want = orc.loglikelihood(cell, fetched[t], zs[p], rs[p])
            assert abs(ll[p, t] - want) <= RTOL * abs(want), (p, t, ll[p, t], want)
    for p in (0, 3, 5):
        one, _ = ctx.eval_datasets(zs[p], rs[p])
        np.testing.assert_allclose(ll[p], one, rtol=1e-13, atol=0)
    part, _ = ctx.eval_datasets_points(zs, rs, 1200, 9100)
    np.testing.assert_array_equal(part, ll[:, 1200:9100])
    buf = ctx.device_alloc(8 * P * T)
    st_dev = ctx.eval_datasets_points_device(buf.ptr, zs, rs)
    assert not st_dev.any()
    np.testing.assert_array_equal(buf.to_host(np.float64, P * T).reshape(P, T), ll)
    buf.free()


@pytest.mark.parametrize('sparse', [1, 0])
def test_c4_million_point_scan(c2, sparse):
    """configs[3] on one GPU: 10^6 parameter points over the C2 model in one call (device planner; non-empty-bin form,
    and every bin visited: non-empty-bin pass + validity pass on the matrix cores); 8 points checked against the oracle, -inf outside the box."""
    from oracle import blueice_oracle as orc
    m, ctx = c2
    P = 10 ** 6
    counts = m.counts()
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(counts)
    z, r = m.random_points(P, seed=31)
    z[123456, 1] = 2.5                        # outside the anchor box
    r[654321, 2] = -0.5                       # unphysical rate
    before = ctx.get_param('n_valid_launches')
    ll, st = ctx.eval(z, r)
    if not sparse:
        assert ctx.get_param('n_valid_launches') == before + 1       # non-empty-bin pass + matrix-core validity pass of every bin
    assert ll[123456] == -np.inf and st[123456] == 1 and ll[654321] == -np.inf and st[654321] == 2
    ok = np.ones(P, bool)
    ok[[123456, 654321]] = False
    assert np.all(np.isfinite(ll[ok])) and not st[ok].any()
    for i in (0, 1, 99999, 250000, 500000, 750001, 999998, P - 1):
        want = orc.loglikelihood(m.cell_model(z[i]), counts, z[i], r[i])
        assert abs(ll[i] - want) <= RTOL * abs(want), (sparse, i, ll[i], want)
    ctx.set_param('sparse', 1)


def test_c5_all_625_anchor_models():
    """configs[4] at its stated size on ONE GPU: 625 anchor models x 6 sources x 50^4 bins (187.5 GB of templates +
    31 GB of MC counts resident in HBM), Beeston-Barlow; one evaluation = one 5.65 GB pass, checked against the
    oracle on the 16 corner models of the point's cell."""
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('C5', bb_source=0)
    ctx = DeviceContext(0)
    if ctx.info()['hbm_bytes'] < 240e9:
        ctx.close()
        pytest.skip("needs 219 GB of HBM")
    m.upload(ctx, threads=14)
    counts = m.counts(dense=True)
    ctx.set_param('sparse', 0)
    ctx.upload_counts(counts)
    z, r = m.random_points(2, seed=2)
    got, st = ctx.eval(z, r)
    assert not st.any()
    want = orc.loglikelihood(m.cell_model(z[0]), counts, z[0], r[0], bb_source=0)
    assert abs(got[0] - want) <= RTOL * abs(want), (got[0], want)
    one, _ = ctx.eval(z[1], r[1])
    assert abs(one[0] - got[1]) <= 1e-13 * abs(one[0])
    plan = ctx.plan(z[:1], r[:1])
    assert plan.bytes == 8 * (16 * 6 + 16 + 1) * m.B
    plan.close()
    ctx.close()


def test_split_scan_validity_pass_small_models():
    """Dense scans over mostly empty data run as non-empty-bin pass + validity pass of every bin on the matrix cores
    (k_scan_valid).  With templates that go NEGATIVE in some bins (a source allowed to be negative, a template with
    negative entries) the validity pass must reproduce scipy's rule -- nan as soon as any bin, empty or not, has
    mu < 0 -- exactly where the oracle says so, and must stay silent elsewhere."""
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    rng = np.random.default_rng(5)
    for S, n_anchor, bins in [(4, (3, 3, 2), (30, 20, 12)), (3, (2, 3), (64, 33)), (4, (), (5000,))]:
        m = SyntheticModel(S, n_anchor, bins, seed=99)
        dense = m.dense_model()
        # source 0 gets a template with a negative dip in a handful of bins of ONE anchor model
        dip = rng.choice(m.B, 7, replace=False)
        flat = dense['ps'].reshape(-1, S, m.B)
        flat[0, 0, dip] = -3.0 * flat[0, 0, dip]
        counts = m.counts(dense=False, scale=0.05 if m.B > 5000 else 0.02)
        counts[dip[0]] = 0.0
        ctx = DeviceContext(0)
        ctx.upload_model(dense['anchor_z'], dense['ps'], dense['mus'])
        ctx.set_allow_negative([0, 1] + [0] * (S - 2))
        ctx.set_param('sparse', 0)
        ctx.upload_counts(counts)
        assert ctx.get_param('split_ready') == 1 and ctx.get_param('compact_ready') == 0
        P = 6000
        z, r = m.random_points(P, seed=3)
        r[::3, 1] = -rng.uniform(0.0, 0.4, len(r[::3]))          # the source that may be negative, sometimes is
        before = ctx.get_param('n_valid_launches')
        got, st = ctx.eval(z, r)
        assert ctx.get_param('n_valid_launches') == before + 1      # the split path took it
        ctx.set_param('scan_split', 0)
        ref_dev, st2 = ctx.eval(z, r)                               # every per-bin term in every bin (k_scan_mfma / k_morph_reduce)
        ctx.set_param('scan_split', 1)
        want = orc.loglikelihood_batch(dense, counts, z[::40], r[::40], allow_negative=[False, True] + [False] * (S - 2))
        np.testing.assert_array_equal(st, st2)
        n_nan = int(np.isnan(ref_dev).sum())
        assert 0 < n_nan < P, n_nan                                  # the case is neither trivial nor all-nan
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref_dev))
        np.testing.assert_array_equal(np.isneginf(got), np.isneginf(ref_dev))
        fin = np.isfinite(ref_dev)
        np.testing.assert_allclose(got[fin], ref_dev[fin], rtol=1e-11)
        for a, b in zip(got[::40], want):
            assert (np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= RTOL * max(1.0, abs(b)), (a, b)
        ctx.close()
