"""Full-size parity at BASELINE.json shapes (C2: 4 src, 5^3 anchors, 100^3 bins) on synthetic tensors:
against the CPU oracle on the same inputs (a handful of points -- 0.25 s each on the host), and through
size-independent properties for the rest."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def c2():
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('C2')
    ctx = DeviceContext(0)
    m.upload(ctx)
    yield m, ctx
    ctx.close()


def test_c2_matches_oracle(c2):
    from oracle import blueice_oracle as orc
    m, ctx = c2
    for dense in (False, True):
        counts = m.counts(dense=dense)
        z0, r0 = m.default_point()
        zs, rs = m.random_points(2, seed=3)
        pts = [(z0, r0), (zs[0], rs[0]), (np.array([-2., 2., 0.]), rs[1])]     # off-grid, random, on-anchor corner
        want = [orc.loglikelihood(m.cell_model(z), counts, z, r) for z, r in pts]
        for sparse in (0, 1):
            ctx.set_param('sparse', sparse)
            ctx.upload_counts(counts)
            # ~1e4 events in 1e6 bins -> the non-empty-bin form is active; 10 events per bin -> it is not
            assert ctx.get_param('compact_ready') == (1 if (sparse and not dense) else 0)
            for (z, r), w in zip(pts, want):
                got, st = ctx.eval(z, r)
                assert st[0] == 0
                assert abs(got[0] - w) <= 1e-10 * max(1.0, abs(w)), (dense, sparse, z, got[0], w)
            toys, tst = ctx.eval_datasets(z0, r0)
            assert abs(toys[0] - want[0]) <= 1e-10 * abs(want[0])


def test_c2_properties(c2):
    m, ctx = c2
    counts = m.counts()
    ctx.set_param('sparse', 0)
    ctx.upload_counts(counts)
    z, r = m.random_points(48, seed=9)
    single = np.array([ctx.eval(z[i], r[i])[0][0] for i in range(8)])
    batch, st = ctx.eval(z, r)
    assert not st.any()
    # batched (cell-grouped, several points per pass) == one at a time
    np.testing.assert_allclose(batch[:8], single, rtol=1e-13)
    # run-to-run bitwise reproducibility (fixed-order reduction, no atomics)
    again, _ = ctx.eval(z, r)
    np.testing.assert_array_equal(batch, again)
    # data-only term: logL(n) - logL(0) at fixed parameters == sum n log mu - sum lgamma(n+1);
    # with all-zero data logL = -sum_s r_s (ps rows are normalised): an exact closed form
    ctx.upload_counts(np.zeros(m.B))
    ll0, _ = ctx.eval(z[:4], r[:4])
    for i in range(4):
        mus = ctx.interpolate('mus', z[i]) * r[i]
        assert abs(ll0[i] + mus.sum()) <= 1e-10 * mus.sum()
    # toy-MC form == per-dataset batched form
    toys = np.stack([m.counts(dataset=t) for t in range(5)])
    ctx.upload_counts(toys)
    a, st = ctx.eval_datasets(z[0], r[0])
    b, _ = ctx.eval(np.tile(z[0], (5, 1)), np.tile(r[0], (5, 1)), dataset=np.arange(5))
    assert st == 0
    np.testing.assert_allclose(a, b, rtol=1e-13)
    # on an anchor the morph is the identity: compare with a d = 0 model made of that anchor alone
    from blueice_amd.device import DeviceContext
    a_idx = m.central_anchor()
    solo = DeviceContext(0)
    solo.begin_model([], m.S, m.B)
    solo.set_anchor(0, m.anchor_ps(a_idx), m.anchor_mus(a_idx))
    solo.end_model()
    solo.upload_counts(counts)
    ctx.upload_counts(counts)
    x, _ = solo.eval(None, r[:1])
    y, _ = ctx.eval(np.array(m.anchor_zs(a_idx)), r[0])
    assert abs(x[0] - y[0]) <= 1e-12 * abs(x[0])
    solo.close()
    # non-empty-bin (sparse) form == dense form on the same batch, and the plan goes stale on re-upload
    ctx.set_param('sparse', 1)
    ctx.upload_counts(counts)
    assert ctx.get_param('compact_ready') == 1
    plan = ctx.plan(z, r)
    assert plan.bytes < 0.05 * 48 * 264e6
    plan.run()
    sp, st = plan.read()
    np.testing.assert_allclose(sp, batch, rtol=1e-12)
    from blueice_amd.exceptions import NotPreparedException
    ctx.upload_counts(counts)
    with pytest.raises(NotPreparedException):
        plan.run()


def test_c5_beeston_barlow_full_bins():
    """BASELINE.json configs[4] at full bin count: Beeston-Barlow, 6 sources, 4 shape parameters, 50^4 bins.
    One grid cell of the anchor grid (2 anchors per axis, 4.8 GB) is enough to exercise every stream of an
    evaluation (16 corners x 5 plain sources + 16 BB-source rows + 16 MC-count rows = 112 rows, 5.65 GB)."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    m = SyntheticModel.named('C5-2anchor', bb_source=0)
    ctx = DeviceContext(0)
    m.upload(ctx)
    counts = m.counts(dense=True)           # ~10 events per bin
    ctx.upload_counts(counts)
    z, r = m.random_points(3, seed=2)
    got, st = ctx.eval(z, r)
    assert not st.any()
    want = orc.loglikelihood(m.cell_model(z[0]), counts, z[0], r[0], bb_source=0)
    assert abs(got[0] - want) <= 1e-10 * max(1.0, abs(want)), (got[0], want)
    one, st1 = ctx.eval(z[1], r[1])
    assert abs(one[0] - got[1]) <= 1e-13 * abs(one[0])
    plan = ctx.plan(z, r)
    assert plan.bytes == 8 * (16 * 6 + 16 + 1) * m.B          # three points of one cell share one pass (G=4)
    ctx.close()


def test_c3_toy_batch(c2):
    """BASELINE.json configs[2] (toy-MC datasets batched per call) at full bin count with a reduced number
    of toys: the dense-count form, the CSR form and the per-dataset point form agree, and one toy is checked
    against the oracle."""
    from oracle import blueice_oracle as orc
    m, ctx = c2
    T = 48
    toys = np.stack([m.counts(dataset=100 + t) for t in range(T)])
    z, r = m.default_point()
    res = {}
    for sparse in (0, 1):
        ctx.set_param('sparse', sparse)
        ctx.upload_counts(toys)
        assert ctx.get_param('csr_ready') == sparse
        res[sparse], st = ctx.eval_datasets(z, r)
        assert st == 0
    np.testing.assert_allclose(res[0], res[1], rtol=1e-13)
    pts, _ = ctx.eval(np.tile(z, (T, 1)), np.tile(r, (T, 1)), dataset=np.arange(T))     # compacted templates
    np.testing.assert_allclose(pts, res[0], rtol=1e-12)
    part, _ = ctx.eval_datasets(z, r, 7, 19)
    np.testing.assert_array_equal(part, res[1][7:19])
    want = orc.loglikelihood(m.cell_model(z), toys[5], z, r)
    assert abs(res[0][5] - want) <= 1e-10 * abs(want)
    ctx.set_param('sparse', 1)


def test_device_planning_equals_host_planning():
    """Large batches are planned on the device (geometry, radix sort by (cell, dataset), item chopping,
    descriptor fill): same numbers as the host planner, including rejected points, several datasets, both
    data forms."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    toys = np.stack([m.counts(dense=False, dataset=t, scale=0.02) for t in range(3)])
    rng = np.random.default_rng(3)
    P = 40000
    z, r = m.random_points(P, seed=12)
    z[::97, 0] = 5.0                      # out of the box
    z[5::131, 1] = np.nan
    r[7::89, 2] = -1.0                    # unphysical
    ds = rng.integers(0, 3, P)
    ds[11::203] = 7                       # bad dataset
    for sparse in (0, 2):
        ctx.set_param('sparse', sparse)
        ctx.upload_counts(toys)
        ctx.set_param('device_plan_min', 0)                 # host planner
        host, hst = ctx.eval(z, r, dataset=ds)
        ctx.set_param('device_plan_min', 1000)              # device planner
        dev, dst = ctx.eval(z, r, dataset=ds)
        np.testing.assert_array_equal(dst, hst)
        assert np.array_equal(np.isneginf(dev), np.isneginf(host)) and np.isneginf(dev).sum() > 800
        fin = np.isfinite(host)
        np.testing.assert_allclose(dev[fin], host[fin], rtol=1e-13)
        plan = ctx.plan(z, r, dataset=ds)
        plan.run()
        again, _ = plan.read()
        np.testing.assert_array_equal(again, dev)
        plan.close()
    # no shape parameters at all (one "cell")
    solo = DeviceContext(0)
    solo.begin_model([], m.S, m.B)
    solo.set_anchor(0, m.anchor_ps(0), m.anchor_mus(0))
    solo.end_model()
    solo.upload_counts(m.counts(dense=True))
    rr = rng.uniform(0.5, 1.5, size=(20000, m.S))
    solo.set_param('device_plan_min', 0)
    a, _ = solo.eval(None, rr)
    solo.set_param('device_plan_min', 1000)
    b, _ = solo.eval(None, rr)
    np.testing.assert_allclose(b, a, rtol=1e-13)
    solo.close()
    ctx.close()


def test_counting_sort_plans_what_the_radix_sort_plans(c2):
    """A scan over one dataset has at most anchors + 1 key values: the device planner orders them with the one-pass counting sort
    (plan_count_sort).  Both sorts are stable, so the plans -- and every result bit -- are the same."""
    m, ctx = c2
    ctx.set_param('sparse', 1)
    ctx.upload_counts(m.counts())
    for P in (1000, 2049, 70001):
        z, r = m.random_points(P, seed=P)
        z[::97, 0] = 5.0                  # rejected points sort behind everything
        r[7::89, 2] = -1.0
        ctx.set_param('device_plan_min', 1)
        ctx.set_param('plan_count_sort', 0)
        a, sa = ctx.eval(z, r)
        ctx.set_param('plan_count_sort', 1)
        b, sb = ctx.eval(z, r)
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(a, b)
        assert np.isneginf(a).sum() >= P // 97
    ctx.set_param('device_plan_min', 1000)


def test_polled_reports_and_stream_synchronisation_agree(c2):
    """The device planner's counters, a plan's status word and the multi-hypothesis toy call's completion come back through pinned
    memory the host polls (poll_result = 1, default); with poll_result = 0 every one of them waits on the stream instead: the same
    plans, the same results."""
    m, ctx = c2
    ctx.set_param('sparse', 1)
    ctx.upload_counts(m.counts())
    z, r = m.random_points(30000, seed=5)
    z[::211, 1] = 9.0
    ctx.set_param('device_plan_min', 1)
    out = {}
    for poll in (1, 0, 1):
        ctx.set_param('poll_result', poll)
        ll, st = ctx.eval(z, r)
        plan = ctx.plan(z[:5000], r[:5000])
        plan.run()
        word = plan.status()
        again, _ = plan.read()
        plan.close()
        out[poll] = (ll, st, word, again)
    ctx.set_param('poll_result', 1)
    ctx.set_param('device_plan_min', 1000)
    for a, b in zip(out[1], out[0]):
        np.testing.assert_array_equal(a, b)
    assert out[1][2] == 1 and np.isneginf(out[1][0]).sum() >= 30000 // 211


def test_matrix_core_scan_kernel_matches_vector_kernel(c2):
    """Scans with many points per grid cell run on the fp64 matrix cores (k_scan_mfma): same numbers as the
    vector kernel (k_morph_reduce) and as the oracle, including -inf / nan bins."""
    from oracle import blueice_oracle as orc
    m, ctx = c2
    counts = m.counts()
    counts[12345] = 2.5           # non-integer count -> -inf for every point of that dataset
    ctx.set_param('sparse', 0)
    ctx.set_param('scan_split', 0)       # this test is about k_scan_mfma itself (per-bin terms in every bin), not the split path
    ctx.upload_counts(counts)
    rng = np.random.default_rng(8)
    n1 = 20000
    z = np.stack([rng.uniform(-0.9, -0.1, n1), rng.uniform(0.1, 0.9, n1), rng.uniform(1.1, 1.9, n1)], 1)   # one cell
    r = rng.uniform(0.5, 1.5, (n1, m.S))
    z = np.concatenate([z, m.random_points(30000, seed=2)[0]])
    r = np.concatenate([r, m.random_points(30000, seed=2)[1]])
    ctx.set_param('scan_mfma', 1)
    before = ctx.get_param('n_scan_launches')
    vec, _ = ctx.eval(z, r)
    assert ctx.get_param('n_scan_launches') == before + 1 and np.all(np.isneginf(vec))
    ctx.set_param('scan_mfma', 0)
    counts[12345] = 3.0
    ctx.upload_counts(counts)
    vec, _ = ctx.eval(z, r)
    ctx.set_param('scan_mfma', 1)
    before = ctx.get_param('n_scan_launches')
    plan = ctx.plan(z, r)
    plan.run()
    mat, st = plan.read()
    plan.close()
    assert ctx.get_param('n_scan_launches') == before + 1          # the matrix-core kernel really ran
    assert not st.any()
    np.testing.assert_allclose(mat, vec, rtol=1e-13)
    for i in (0, 700, 19999):
        want = orc.loglikelihood(m.cell_model(z[i]), counts, z[i], r[i])
        assert abs(mat[i] - want) <= 1e-10 * abs(want)
    counts[777] = np.nan
    ctx.upload_counts(counts)
    before = ctx.get_param('n_scan_launches')
    bad, _ = ctx.eval(z[:n1], r[:n1])
    assert ctx.get_param('n_scan_launches') == before + 1
    assert np.all(np.isnan(bad))
    ctx.set_param('scan_split', 1)
    ctx.set_param('sparse', 1)


@pytest.mark.parametrize('S,n_anchor,bins', [
    (3, (), (2000,)),              # no shape parameter: 3 streams  (1 K group, masked)
    (4, (), (1500,)),              # 4 streams                       (1 K group)
    (3, (3,), (40, 30)),           # 6 streams                       (2 K groups, masked)
    (4, (2,), (1100,)),            # 8 streams                       (2 K groups)
    (3, (3, 2), (37, 41)),         # 12 streams                      (4 K groups, masked)
    (4, (2, 3), (50, 21)),         # 16 streams                      (4 K groups)
    (5, (2, 2, 3), (13, 11, 9)),   # 40 streams: beyond the kernel's 32 -> the vector kernel takes it
    (3, (2, 3, 2), (12, 10, 9)),   # 24 streams                      (8 K groups, masked)
])
def test_matrix_core_scan_small_stream_counts(S, n_anchor, bins):
    """Every (K groups, padding mask, strip width) variant of k_scan_mfma against the oracle on small models."""
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel(S, n_anchor, bins, seed=77)
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_param('sparse', 0)
    ctx.set_param('scan_split', 0)       # k_scan_mfma itself; the split path has its own test (test_fullsize_reference_gpu.py)
    model = m.dense_model()
    z, r = m.random_points(3000, seed=5)
    for dense in (False, True):
        counts = m.counts(dense=dense)
        ctx.upload_counts(counts)
        want = np.array([orc.loglikelihood(model, counts, z[i], r[i]) for i in range(0, len(z), 50)])
        for cb in (2, 4):
            ctx.set_param('scan_cb', cb)
            before = ctx.get_param('n_scan_launches')
            got, st = ctx.eval(z, r)
            ran = ctx.get_param('n_scan_launches') - before
            assert ran == (1 if S * 2 ** len(n_anchor) <= 32 else 0)
            assert not st.any()
            np.testing.assert_allclose(got[::50], want, rtol=1e-10)
            ctx.set_param('scan_mfma', 0)
            vec, _ = ctx.eval(z, r)
            ctx.set_param('scan_mfma', 1)
            np.testing.assert_allclose(got, vec, rtol=1e-12)
    ctx.close()
