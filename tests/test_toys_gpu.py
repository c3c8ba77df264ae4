"""Device-side toy-MC generation (bi_generate_toys): statistical equivalence with Poisson(mu_b) per bin,
reproducibility, and consistency of the likelihood of generated toys with the uploaded-counts path."""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu


def make_ctx(mu):
    """d = 0, one source whose expected counts per bin are exactly `mu`."""
    from blueice_amd.device import DeviceContext
    ctx = DeviceContext(0)
    tot = mu.sum()
    ctx.upload_model([], (mu / tot)[None, :], np.array([tot]))
    return ctx


def test_poisson_statistics_small_and_large_means():
    mu = np.concatenate([np.geomspace(0.004, 9.5, 40), np.linspace(10., 80., 24), [0.0, 250.0]])
    ctx = make_ctx(mu)
    T = 6000
    ctx.generate_toys(None, None, T, seed=12345)
    toys = np.stack([ctx.download_counts(t) for t in range(T)])
    assert np.all(toys == np.floor(toys)) and np.all(toys >= 0)
    assert np.all(toys[:, -2] == 0)                              # mu = 0 -> never an event
    mean, var = toys.mean(axis=0), toys.var(axis=0, ddof=1)
    se = np.sqrt(np.maximum(mu, 1e-12) / T)
    assert np.all(np.abs(mean - mu) <= 5 * se + 1e-12), np.max(np.abs(mean - mu) / (se + 1e-12))
    ok = mu > 0.5
    assert np.all(np.abs(var[ok] / mu[ok] - 1) < 6 * np.sqrt(2.0 / T) + 3.0 / (mu[ok] * np.sqrt(T)))
    # frequencies of n = 0, 1, 2 against the pmf (both sampler branches)
    for b in (5, 25, 39, 45, 60):
        for k in (0, 1, 2, int(mu[b])):
            p = stats.poisson(mu[b]).pmf(k)
            f = np.mean(toys[:, b] == k)
            assert abs(f - p) <= 5 * np.sqrt(p * (1 - p) / T) + 1e-4, (b, k, f, p)
    # neighbouring bins / datasets are uncorrelated
    c = np.corrcoef(toys[:, 50], toys[:, 51])[0, 1]
    assert abs(c) < 5 / np.sqrt(T)
    # chi-square of the totals: sum over bins ~ Poisson(sum mu)
    tot = toys.sum(axis=1)
    assert abs(tot.mean() - mu.sum()) < 5 * np.sqrt(mu.sum() / T)
    # reproducible, seed-dependent
    ctx.generate_toys(None, None, 50, seed=12345)
    again = np.stack([ctx.download_counts(t) for t in range(50)])
    np.testing.assert_array_equal(again, toys[:50])
    ctx.generate_toys(None, None, 50, seed=12346)
    other = np.stack([ctx.download_counts(t) for t in range(50)])
    assert not np.array_equal(other, toys[:50])
    ctx.close()


def test_generated_toys_evaluate_like_uploaded_counts():
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    z, r = m.default_point()
    T = 40
    ctx.generate_toys(z, 0.02 * r, T, seed=7)          # ~200 events in 4420 bins
    assert ctx.get_param('csr_ready') == 1 and ctx.get_param('compact_ready') == 1
    gen, st = ctx.eval_datasets(z, 0.02 * r)
    pts, _ = ctx.eval(np.tile(z, (T, 1)), np.tile(0.02 * r, (T, 1)), dataset=np.arange(T))
    np.testing.assert_allclose(pts, gen, rtol=1e-12)
    dense = np.stack([ctx.download_counts(t) for t in range(T)])
    assert 100 < dense.sum(axis=1).mean() < 400
    want = orc.loglikelihood(m.dense_model(), dense[3], z, 0.02 * r)
    assert abs(gen[3] - want) <= 1e-10 * abs(want)
    ctx.upload_counts(dense)
    up, _ = ctx.eval_datasets(z, 0.02 * r)
    np.testing.assert_allclose(up, gen, rtol=1e-13)
    # a toy set too big for the compaction budget still serves the toy-MC form; point calls say why not
    ctx.set_param('compact_budget', 1024)
    ctx.generate_toys(z, 0.02 * r, T, seed=7)
    assert ctx.get_param('compact_ready') == 0
    again, _ = ctx.eval_datasets(z, 0.02 * r)
    np.testing.assert_array_equal(again, gen)
    from blueice_amd.exceptions import NotPreparedException
    with pytest.raises(NotPreparedException):
        ctx.eval(z, r)
    ctx.close()


def test_simulate_toys_through_the_likelihood_class():
    import model_zoo
    ns = model_zoo.namespace_of('blueice_amd')
    lf, _, _ = model_zoo.d3_small(ns)
    lf.simulate_toys(300, seed=5, shift=0.2, stretch=-0.3, tilt=0.5, s1_rate_multiplier=2.0)
    ll = lf.eval_toys(shift=0.2, stretch=-0.3, tilt=0.5, s1_rate_multiplier=2.0)
    assert ll.shape == (300,) and np.all(np.isfinite(ll))
    assert abs(lf(shift=0.2, stretch=-0.3, tilt=0.5, s1_rate_multiplier=2.0) - ll[0]) <= 1e-13 * abs(ll[0])
    mus = lf.ctx.interpolate('mus', [0.2, -0.3, 0.5]) * np.array([1., 2., 1., 1.])
    n_tot = np.array([lf.ctx.download_counts(t).sum() for t in range(300)])
    assert abs(n_tot.mean() - mus.sum()) < 5 * np.sqrt(mus.sum() / 300)
    # the truth should, on average, be favoured over a displaced hypothesis
    other = lf.eval_toys(shift=-0.8, stretch=0.7, tilt=-0.5, s1_rate_multiplier=0.5)
    assert np.mean(ll - other) > 0


def test_uploaded_dataset_batches_tiled_and_row_kernels():
    """bi_eval_datasets over UPLOADED sparse datasets large enough for the tiled kernel (bin tiles of log mu staged through
    LDS over tile-major 4-byte entries): every dataset against the oracle; a count beyond the 19 bits of an entry, a
    non-integer and a nan count send the whole batch back to the row kernel, with scipy's values for those datasets."""
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel(2, (3,), (64, 32, 32), seed=5)                 # 65 536 bins = 8 tiles
    ctx = DeviceContext(0)
    try:
        m.upload(ctx)
        rng = np.random.default_rng(6)
        T, B = 80, m.B
        counts = np.zeros((T, B))
        for t in range(T):
            hit = rng.choice(B, size=int(rng.integers(1500, 3000)), replace=False)
            counts[t, hit] = rng.integers(1, 6, size=len(hit))
        counts[7, :] = 0                                              # an empty dataset in the middle
        counts[11, 123] = 524287.                                     # the largest count an entry holds
        z, r = m.random_points(1, seed=3)
        z, r = z[0], r[0]
        dense = m.dense_model()
        ctx.set_param('sparse', 1)
        ctx.upload_counts(counts)
        ll, st = ctx.eval_datasets(z, r)
        assert st == 0
        for t in (0, 7, 11, 40, T - 1):
            want = orc.loglikelihood(dense, counts[t], z, r)
            assert abs(ll[t] - want) <= 1e-10 * max(1.0, abs(want)), (t, ll[t], want)
        ctx.set_param('dot_tiled', 0)
        rows, _ = ctx.eval_datasets(z, r)
        ctx.set_param('dot_tiled', 1)
        np.testing.assert_allclose(rows, ll, rtol=1e-13, atol=0)
        counts[11, 123] = 524288.                                     # one more: does not fit
        counts[12, 77] = 2.5                                          # scipy: -inf
        counts[13, 5] = np.nan                                        # scipy: nan
        ctx.upload_counts(counts)
        ll2, st = ctx.eval_datasets(z, r)
        assert ll2[12] == -np.inf and np.isnan(ll2[13])
        want = orc.loglikelihood(dense, counts[11], z, r)
        assert abs(ll2[11] - want) <= 1e-10 * abs(want)
        keep = np.ones(T, bool)
        keep[[11, 12, 13]] = False
        np.testing.assert_array_equal(ll2[keep], rows[keep])          # the row kernel again: same bits as before
    finally:
        ctx.close()


@pytest.mark.parametrize('seed', range(6))
def test_tiled_dataset_kernel_random_shapes(seed):
    """The tiled kernel against the row kernel over random bin counts (last tile partial), dataset counts, fill levels,
    empty datasets, datasets living in a single tile, and sub-ranges of datasets."""
    from blueice_amd.device import DeviceContext
    rng = np.random.default_rng(700 + seed)
    B = int(rng.integers(4 * 8192, 9 * 8192)) + int(rng.integers(0, 5000))
    T = int(rng.integers(64, 160))
    mu = rng.random(B) ** 2 * 3 + 0.05
    ctx = make_ctx(mu)
    try:
        counts = np.zeros((T, B))
        for t in range(T):
            k = int(rng.integers(3000, 9000))
            hit = rng.choice(B, size=k, replace=False)
            counts[t, hit] = rng.integers(1, 40, size=k)
        counts[int(rng.integers(T))] = 0
        one_tile = int(rng.integers(T))
        counts[one_tile] = 0
        counts[one_tile, 8192 * 2 + rng.choice(8192, 500, replace=False)] = 3
        ctx.set_param('sparse', 1)
        ctx.upload_counts(counts)
        tiled, st = ctx.eval_datasets(None, [1.3])
        assert st == 0 and np.all(np.isfinite(tiled))
        ctx.set_param('dot_tiled', 0)
        rows, _ = ctx.eval_datasets(None, [1.3])
        ctx.set_param('dot_tiled', 1)
        np.testing.assert_allclose(tiled, rows, rtol=1e-13, atol=0)
        assert ctx.get_param('dot_lanes') == 0                        # (the default: by entry width -- here 8 lanes per run, 96 entry slots)
        ctx.set_param('dot_lanes', 16)                                # round 3's shape: 16 lanes, 128 slots
        wide, _ = ctx.eval_datasets(None, [1.3])
        ctx.set_param('dot_lanes', 0)
        np.testing.assert_allclose(wide, rows, rtol=1e-13, atol=0)
        # round 4's short call: descriptors in the kernel arguments (bit 1), the finish with 64 datasets per block (2), the
        # completion word polled instead of a stream synchronise (4) -- the same numbers whichever way the call travels;
        # only the finish changes the order of a dataset's ~n_tiles additions
        assert ctx.get_param('toy_fast_call') == 7
        polled = ctx.get_param('n_toy_polled')
        again, _ = ctx.eval_datasets(None, [1.3])
        np.testing.assert_array_equal(again, tiled)
        assert ctx.get_param('n_toy_polled') == polled + 1
        by_bits = {}
        for bits in (0, 1, 2, 3, 6):
            ctx.set_param('toy_fast_call', bits)
            by_bits[bits], st_b = ctx.eval_datasets(None, [1.3])
            assert st_b == 0
        ctx.set_param('toy_fast_call', 7)
        np.testing.assert_array_equal(by_bits[1], by_bits[0])
        np.testing.assert_array_equal(by_bits[3], by_bits[2])
        np.testing.assert_array_equal(by_bits[6], by_bits[2])
        np.testing.assert_array_equal(tiled, by_bits[2])
        np.testing.assert_allclose(by_bits[2], by_bits[0], rtol=1e-13, atol=0)
        for k in range(300):                                          # (past a 256th call: those synchronise the stream as well)
            more, _ = ctx.eval_datasets(None, [1.3 + 0.001 * (k % 3)])
        more, _ = ctx.eval_datasets(None, [1.3])
        np.testing.assert_array_equal(more, tiled)
        lo, hi = sorted(rng.choice(T - 64, 2, replace=False))
        part, _ = ctx.eval_datasets(None, [1.3], int(lo), int(hi) + 64)
        np.testing.assert_array_equal(part, tiled[lo:hi + 64])
        want = np.array([np.sum(stats.poisson(1.3 * mu).logpmf(counts[t])) for t in (0, one_tile, T - 1)])
        np.testing.assert_allclose(tiled[[0, one_tile, T - 1]], want, rtol=1e-10)
        assert ctx.get_param('tm_entry_bytes') == 4                   # (counts up to 39: four-byte entries)
        # counts of at most 7 everywhere (what toys of sparse expectations look like): two-byte entries, eight per load --
        # against the row kernel, against four-byte entries of the same data, and with 4 lanes per run instead of 8
        small = np.minimum(counts, rng.integers(1, 8, size=counts.shape))
        ctx.upload_counts(small)
        two, st2 = ctx.eval_datasets(None, [1.3])
        assert st2 == 0 and ctx.get_param('tm_entry_bytes') == 2
        ctx.set_param('dot_tiled', 0)
        rows2, _ = ctx.eval_datasets(None, [1.3])
        ctx.set_param('dot_tiled', 1)
        np.testing.assert_allclose(two, rows2, rtol=1e-13, atol=0)
        ctx.set_param('dot_lanes', 8)                                 # (the default for two-byte entries is 4 lanes per run)
        eight, _ = ctx.eval_datasets(None, [1.3])
        ctx.set_param('dot_lanes', 0)
        np.testing.assert_allclose(eight, rows2, rtol=1e-13, atol=0)
        ctx.set_param('dot_entry16', 0)
        four, _ = ctx.eval_datasets(None, [1.3])
        assert ctx.get_param('tm_entry_bytes') == 4
        ctx.set_param('dot_entry16', 1)
        np.testing.assert_allclose(four, rows2, rtol=1e-13, atol=0)
        part2, _ = ctx.eval_datasets(None, [1.3], int(lo), int(hi) + 64)
        assert ctx.get_param('tm_entry_bytes') == 2
        np.testing.assert_array_equal(part2, two[lo:hi + 64])
        want2 = np.array([np.sum(stats.poisson(1.3 * mu).logpmf(small[t])) for t in (0, one_tile, T - 1)])
        np.testing.assert_allclose(two[[0, one_tile, T - 1]], want2, rtol=1e-10)
    finally:
        ctx.close()


def test_event_by_event_generation_of_sparse_toys():
    """Toys of sparse expectations are drawn event by event (N ~ Poisson(sum mu), bins by bisection in the cumulative
    sums, sorted and run-length encoded per toy): the same law as one Poisson draw per bin.  Checked: totals, per-bin
    and per-group means and variances, zero fractions, sorted unique bin lists, reproducibility and the dataset-number
    semantics of toy_offset, the data-only lgamma term through the likelihood, and the per-bin generator on request."""
    rng = np.random.default_rng(41)
    B = 40000
    mu = np.where(rng.random(B) < 0.3, 0.0, rng.random(B) ** 4 * 0.2)      # many exact zeros, sum ~ 1100
    mu[:40] = np.linspace(0.5, 6.0, 40)                                     # a few busy bins
    M = mu.sum()
    ctx = make_ctx(mu)
    try:
        T = 5000
        ctx.set_param('sparse', 1)
        ctx.generate_toys(None, None, T, seed=99)
        assert ctx.get_param('last_toy_method') == 1
        toys = np.stack([ctx.download_counts(t) for t in range(0, T, 5)])   # 1000 of them, densely
        assert np.all(toys == np.floor(toys)) and np.all(toys >= 0) and np.all(toys[:, mu == 0] == 0)
        tot = toys.sum(axis=1)
        assert abs(tot.mean() - M) < 5 * np.sqrt(M / len(toys)) and abs(tot.var(ddof=1) / M - 1) < 6 * np.sqrt(2 / len(toys))
        busy = toys[:, :40]
        se = np.sqrt(mu[:40] / len(toys))
        assert np.all(np.abs(busy.mean(axis=0) - mu[:40]) < 5 * se)
        assert np.all(np.abs(busy.var(axis=0, ddof=1) / mu[:40] - 1) < 6 * np.sqrt(2 / len(toys)) + 3 / (mu[:40] * np.sqrt(len(toys))))
        groups = np.array_split(np.arange(40, B), 60)                        # the sparse bulk, in 60 groups of bins
        for gidx in groups:
            m = mu[gidx].sum()
            x = toys[:, gidx].sum(axis=1)
            assert abs(x.mean() - m) < 5 * np.sqrt(m / len(toys)) + 1e-9
            assert abs(x.var(ddof=1) / m - 1) < 6 * np.sqrt(2 / len(toys)) + 3 / (m * np.sqrt(len(toys)))
        for b in (0, 10, 39):
            p = np.exp(-mu[b])
            assert abs(np.mean(toys[:, b] == 0) - p) < 5 * np.sqrt(p * (1 - p) / len(toys)) + 1e-3
        c = np.corrcoef(toys[:, 5], toys[:, 6])[0, 1]
        assert abs(c) < 5 / np.sqrt(len(toys))
        # the likelihood of the toys (data-only lgamma term included) against scipy on the fetched counts
        ll, st = ctx.eval_datasets(None, [1.1])
        assert st == 0
        for k, t in enumerate(range(0, 50, 5)):
            want = np.sum(stats.poisson(1.1 * mu).logpmf(toys[k]))
            assert abs(ll[t] - want) <= 1e-10 * abs(want)
        # same seed -> same toys; a range drawn with toy_offset is that range of the ensemble
        first = ctx.download_counts(3), ctx.download_counts(T - 1)
        ctx.generate_toys(None, None, T, seed=99)
        np.testing.assert_array_equal(ctx.download_counts(3), first[0])
        ctx.set_param('toy_offset', T - 10)
        ctx.generate_toys(None, None, 10, seed=99)
        np.testing.assert_array_equal(ctx.download_counts(9), first[1])
        ctx.set_param('toy_offset', 0)
        ctx.generate_toys(None, None, 50, seed=100)
        assert np.any(ctx.download_counts(3) != first[0])
        # bin by bin on request: another stream of random numbers, the same statistics
        ctx.set_param('toy_events', 0)
        ctx.generate_toys(None, None, 1000, seed=99)
        assert ctx.get_param('last_toy_method') == 0
        per_bin = np.stack([ctx.download_counts(t) for t in range(1000)])
        assert abs(per_bin.sum(axis=1).mean() - M) < 5 * np.sqrt(M / 1000)
        ctx.set_param('toy_events', 1)
    finally:
        ctx.close()


def test_event_by_event_generation_with_very_few_events():
    """sum mu = 2.5 over 8 000 bins: N comes from the inversion sampler, most toys have a handful of events or none."""
    rng = np.random.default_rng(43)
    B = 8000
    mu = np.zeros(B)
    live = rng.choice(B, 400, replace=False)
    mu[live] = rng.random(400)
    mu *= 2.5 / mu.sum()
    ctx = make_ctx(mu)
    try:
        T = 20000
        ctx.set_param('sparse', 1)
        ctx.generate_toys(None, None, T, seed=5)
        assert ctx.get_param('last_toy_method') == 1
        nnz = ctx.get_param('nnz_total')
        toys = np.stack([ctx.download_counts(t) for t in range(0, T, 10)])
        tot = toys.sum(axis=1)
        assert np.all(toys[:, mu == 0] == 0)
        assert abs(tot.mean() - 2.5) < 5 * np.sqrt(2.5 / len(toys)) and abs(np.mean(tot == 0) - np.exp(-2.5)) < 0.03
        assert abs(nnz / T - 2.5) < 0.1                         # (hardly any bin is hit twice)
        ll, st = ctx.eval_datasets(None, [0.8])
        assert st == 0
        for k in (0, 1, 2, 3):
            want = np.sum(stats.poisson(0.8 * mu).logpmf(toys[k]))
            assert abs(ll[10 * k] - want) <= 1e-10 * max(1.0, abs(want))
    finally:
        ctx.close()


def test_event_by_event_generation_with_many_events_per_toy():
    """sum mu = 20 000 over 200 000 bins: more than 16 384 events per toy, the size from which the per-toy sort is the
    bitonic network instead of the radix sort."""
    rng = np.random.default_rng(47)
    B = 200000
    mu = rng.random(B) * 0.2
    mu *= 20000.0 / mu.sum()
    ctx = make_ctx(mu)
    try:
        T = 300
        ctx.set_param('sparse', 1)
        ctx.generate_toys(None, None, T, seed=11)
        assert ctx.get_param('last_toy_method') == 1
        toys = np.stack([ctx.download_counts(t) for t in range(T)])
        tot = toys.sum(axis=1)
        assert abs(tot.mean() - 20000) < 5 * np.sqrt(20000 / T) and 0.7 < tot.var(ddof=1) / 20000 < 1.4
        for gidx in np.array_split(np.arange(B), 40):
            m = mu[gidx].sum()
            x = toys[:, gidx].sum(axis=1)
            assert abs(x.mean() - m) < 5 * np.sqrt(m / T)
        ll, st = ctx.eval_datasets(None, [0.9])
        want = np.sum(stats.poisson(0.9 * mu).logpmf(toys[7]))
        assert st == 0 and abs(ll[7] - want) <= 1e-10 * abs(want)
        again = ctx.download_counts(5).copy()
        ctx.generate_toys(None, None, 10, seed=11)
        np.testing.assert_array_equal(ctx.download_counts(5), again)
    finally:
        ctx.close()
