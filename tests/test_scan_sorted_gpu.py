"""Scans over DENSE data on the matrix cores run on a copy of the rows whose bins are ordered by their count, where
sum n log mu over a lane's bins is n log of their product (k_scan_mfma PROD = 2, ensure_sorted_rows; VERDICT round 2
item 3).  Every bin is still visited and every scipy edge case must come out as before: against the bin-order kernel
(scan_pow = 0), the single-point kernel and the oracle.  Also the product form of the sparse path (PROD = 1) with
expectations whose intermediate products go subnormal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def mini():
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_param('sparse', 0)
    yield m, ctx
    ctx.close()


def _scan(ctx, z, r, pow_on):
    ctx.set_param('scan_pow', pow_on)
    before = ctx.get_param('n_scan_launches'), ctx.get_param('n_sorted_scans')
    ll, st = ctx.eval(z, r)
    after = ctx.get_param('n_scan_launches'), ctx.get_param('n_sorted_scans')
    assert after[0] == before[0] + 1, 'the batch did not take the matrix-core scan kernel'
    assert after[1] == before[1] + (1 if pow_on else 0)
    return ll, st


def test_sorted_rows_scan_equals_bin_order_scan_and_oracle(mini):
    from oracle import blueice_oracle as orc
    m, ctx = mini
    counts = m.counts(dense=True)
    assert (counts > 0).mean() > 0.99
    ctx.upload_counts(counts)
    z, r = m.random_points(3000, seed=21)
    a, st_a = _scan(ctx, z, r, 1)
    b, st_b = _scan(ctx, z, r, 0)
    assert not st_a.any() and not st_b.any()
    np.testing.assert_allclose(a, b, rtol=1e-13)
    dense = m.dense_model()
    for i in range(0, 3000, 331):
        want = orc.loglikelihood(dense, counts, z[i], r[i])
        assert abs(a[i] - want) <= 1e-10 * abs(want)
        one, _ = ctx.eval(z[i], r[i])
        assert abs(one[0] - a[i]) <= 1e-12 * abs(want)
    # a second upload of data rebuilds the sorted copy for the new counts
    counts2 = m.counts(dense=True, dataset=3)
    ctx.upload_counts(counts2)
    a2, _ = _scan(ctx, z, r, 1)
    want = orc.loglikelihood(dense, counts2, z[7], r[7])
    assert abs(a2[7] - want) <= 1e-10 * abs(want)


def test_sorted_rows_scan_keeps_scipys_edge_cases(mini):
    """nan / negative / non-integer counts, empty bins, a source allowed negative that drives some bins below zero:
    the same nan / -inf pattern and the same numbers as the bin-order kernel."""
    m, ctx = mini
    counts = m.counts(dense=True)
    counts[5] = np.nan
    z, r = m.random_points(2048, seed=22)
    for variant in range(4):
        c = m.counts(dense=True)
        if variant == 0:
            c[100:140] = 0.0                       # a run of empty bins among the data
        elif variant == 1:
            c[17] = 2.5                            # not a count: -inf everywhere
        elif variant == 2:
            c[4000] = -1.0
        else:
            c[33] = np.nan
        ctx.upload_counts(c)
        a, _ = _scan(ctx, z, r, 1)
        b, _ = _scan(ctx, z, r, 0)
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        np.testing.assert_array_equal(np.isneginf(a), np.isneginf(b))
        ok = np.isfinite(b)
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-13)
        assert ok.all() == (variant == 0)
    # negative expectations: source 1 may go negative, and does for some points
    ctx.set_allow_negative([0, 1, 0, 0])
    c = m.counts(dense=True)
    ctx.upload_counts(c)
    r2 = r.copy()
    r2[::3, 1] = -3.5
    a, _ = _scan(ctx, z, r2, 1)
    b, _ = _scan(ctx, z, r2, 0)
    assert np.isnan(b).any() and np.isfinite(b).any()
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    ok = np.isfinite(b)
    np.testing.assert_allclose(a[ok], b[ok], rtol=1e-13)


def test_products_that_leave_the_double_range_take_the_bin_wise_form(mini):
    """Rates of 1e45: every mu is ~1e47, a product of eight overflows -- the kernel must notice and fall back."""
    from oracle import blueice_oracle as orc
    m, ctx = mini
    counts = m.counts(dense=True)
    ctx.upload_counts(counts)
    z, r = m.random_points(2048, seed=23)
    r = r * 1e45
    a, st = _scan(ctx, z, r, 1)
    assert not st.any() and np.all(np.isfinite(a))
    dense = m.dense_model()
    for i in (0, 1000, 2047):
        want = orc.loglikelihood(dense, counts, z[i], r[i])
        assert abs(a[i] - want) <= 1e-10 * abs(want)
    tiny = r * 1e-90 * 1e-45                               # and the other way: mu ~ 1e-43, products of eight underflow
    a, st = _scan(ctx, z, tiny, 1)
    for i in (0, 1000, 2047):
        want = orc.loglikelihood(dense, counts, z[i], tiny[i])
        assert abs(a[i] - want) <= 1e-10 * abs(want)


def test_sparse_product_form_with_subnormal_intermediates():
    """Non-empty-bin form (PROD = 1): counts of 1 and 2 whose expectations are ~1e-80 next to ~1e+3 -- a pair product of
    the small ones is subnormal while the product of all four is a normal number again (ADVICE round 2,
    bi_scan_sorted.h: the intermediates are checked too)."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(5)
    B, S = 8192, 2
    anchor_z = [np.array([0.0, 1.0])]
    ps = rng.uniform(0.5, 1.5, size=(2, S, B))
    hot = np.arange(0, 512)                                 # the bins with data
    ps[:, :, hot[0::2]] *= 1e-165                          # alternating 1e-162 and 1e3 expectations: (1e-162)^2 is subnormal,
    ps[:, :, hot[1::2]] *= 1e+3                            # times (1e3)^2 ... stays tiny but the pair products differ wildly
    mus = np.full((2, S), 1e3)
    counts = np.zeros(B)
    counts[hot] = rng.integers(1, 3, size=len(hot))
    model = dict(anchor_z=anchor_z, ps=ps, mus=mus, n_model=None)
    ctx = DeviceContext(0)
    ctx.set_param('sparse', 2)
    ctx.upload_model(anchor_z, ps, mus)
    ctx.upload_counts(counts)
    assert ctx.get_param('compact_ready') == 1
    P = 1024
    z = rng.uniform(0, 1, size=(P, 1))
    r = rng.uniform(0.5, 1.5, size=(P, S))
    before = ctx.get_param('n_scan_launches')
    ll, st = ctx.eval(z, r)
    assert ctx.get_param('n_scan_launches') == before + 1
    assert not st.any()
    for i in range(0, P, 97):
        want = orc.loglikelihood(model, counts, z[i], r[i])
        assert np.isfinite(want) and abs(ll[i] - want) <= 1e-12 * abs(want), (i, ll[i], want)
    ctx.close()


_FIRST, _COUNT = (int(v) for v in __import__('os').environ.get('BLUEICE_FUZZ_SEEDS', '0:0').split(':'))


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_sorted_rows_random_models_match_oracle(seed):
    """Random small models (0-3 shape axes, ragged bin counts, exact zeros in the templates), ONE dense dataset with the
    odd count thrown in, scans large enough for the matrix-core kernel: count-sorted rows against the oracle point by
    point, nan / -inf pattern included.  BLUEICE_FUZZ_SEEDS=first:count widens the campaign."""
    from test_fuzz_gpu import random_case, random_points
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(8000 + seed)
    ctx = DeviceContext(0)
    ctx.set_param('sparse', 0)
    used = 0
    for rep in range(3):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([1, 2, 3, 4, 5, 8]))
        B = int(rng.choice([64, 100, 511, 512, 700, 1300]))
        model, _ = random_case(rng, d, S, B, -1)
        lam = (model['mus'].reshape(-1, S)[0][:, None] * model['ps'].reshape(-1, S, B)[0]).sum(axis=0)
        counts = rng.poisson(np.maximum(lam, 0.02) * (12.0 / max(lam.mean(), 1e-9))).astype(float)      # ~12 events per bin
        if rng.random() < 0.4:
            counts[rng.integers(B)] = rng.choice([np.nan, -1.0, 2.5])
        if rng.random() < 0.3:
            counts[rng.integers(0, B, 5)] = 0.0
        P = int(rng.integers(700, 1600))
        z, r = random_points(rng, model, P, S)
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
        ctx.upload_counts(counts)
        before = ctx.get_param('n_sorted_scans')
        got, st = ctx.eval(z if d else None, r)
        used += ctx.get_param('n_sorted_scans') > before
        for i in range(0, P, 7):
            w = orc.loglikelihood(model, counts, z[i], r[i])
            g = got[i]
            assert (np.isnan(w) and np.isnan(g)) or g == w or (np.isfinite(w) and abs(g - w) <= 1e-10 * max(1, abs(w))), \
                (seed, rep, d, S, B, i, g, w, st[i])
    assert used >= 1 or _COUNT
    ctx.close()


def test_every_stream_count_from_1_to_32():
    """The count-ordered scan kernel has one variant per number of 4-stream groups (1..8), with or without padding streams
    in the last group: every NS = sources x corners from 1 to 32 (no shape axis: NS = sources; one axis: NS = 2 sources),
    dense data and the compacted non-empty bins of sparse data, against the oracle."""
    from test_fuzz_gpu import random_case, random_points
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(4321)
    ctx = DeviceContext(0)
    seen = set()
    for NS in range(1, 33):
        d = 1 if (NS % 2 == 0 and NS % 3 == 0) else 0           # (some of the even counts as two corners x NS / 2 sources)
        S = NS // 2 if d else NS
        B = int(rng.choice([192, 512, 700]))
        model, _ = random_case(rng, d, S, B, -1)
        if d:                                                   # at least two anchors on the axis: two corners per source
            while len(model['anchor_z'][0]) < 2:
                model, _ = random_case(rng, d, S, B, -1)
        lam = (model['mus'].reshape(-1, S)[0][:, None] * model['ps'].reshape(-1, S, B)[0]).sum(axis=0)
        P = 900
        z, r = random_points(rng, model, P, S)
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
        for sparse in (0, 1):
            ctx.set_param('sparse', sparse)
            scale = 12.0 if not sparse else 0.4
            counts = rng.poisson(np.maximum(lam, 0.02) * (scale / max(lam.mean(), 1e-9))).astype(float)
            ctx.upload_counts(counts)
            before = ctx.get_param('n_scan_launches')
            got, st = ctx.eval(z if d else None, r)
            if ctx.get_param('n_scan_launches') > before:
                seen.add((NS, sparse))
            for i in range(0, P, 11):
                w = orc.loglikelihood(model, counts, z[i], r[i])
                g = got[i]
                assert (np.isnan(w) and np.isnan(g)) or g == w or (np.isfinite(w) and abs(g - w) <= 1e-10 * max(1, abs(w))), \
                    (NS, sparse, i, g, w, st[i])
    ctx.close()
    # the matrix-core scan ran for every stream count in at least one data form
    assert {ns for ns, _ in seen} == set(range(1, 33)), sorted(set(range(1, 33)) - {ns for ns, _ in seen})


@pytest.mark.parametrize('sparse', [0, 1])
def test_mixed_strips_shared_among_the_waves_of_a_cell(mini, sparse):
    """Strips whose 64 bins do not carry one count -- two runs meeting, the end of the data next to the padding, odd counts --
    are worked by all waves of a cell together (scan_share_slow, the default) instead of by the wave that owns the strip:
    the same likelihoods (the partial sums are added in another order: 1e-12), edge cases included, for dense data and for
    the compacted non-empty bins, and for few points per cell (fewer work items than waves)."""
    from oracle import blueice_oracle as orc
    m, ctx = mini
    rng = np.random.default_rng(5)
    counts = m.counts(dense=not sparse).astype(float)
    if not sparse:
        counts[rng.integers(0, len(counts), 7)] = [np.nan, -1.0, 2.5, 0.0, 0.0, 33.0, 1e9]      # odd strips in the middle of runs
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(counts)
    dense = m.dense_model()
    for P in (3000, 1500):
        z, r = m.random_points(P, seed=40 + P)
        out = {}
        for share in (1, 0):
            ctx.set_param('scan_share_slow', share)
            before = ctx.get_param('n_scan_launches')
            out[share], st = ctx.eval(z, r)
            assert ctx.get_param('n_scan_launches') == before + 1, 'the batch did not take the matrix-core scan kernel'
        ctx.set_param('scan_share_slow', 1)
        a, b = out[1], out[0]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b))
        ok = np.isfinite(a)
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-12)
        for i in range(0, P, 97):
            want = orc.loglikelihood(dense, counts, z[i], r[i])
            assert (np.isnan(want) and np.isnan(a[i])) or a[i] == want or abs(a[i] - want) <= 1e-10 * max(1.0, abs(want)), (P, i, a[i], want)
