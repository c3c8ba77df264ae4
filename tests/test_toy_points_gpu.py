"""The toy-MC form over several parameter points (bi_eval_datasets_points, csrc/bi_toy_points.h): P hypotheses x T datasets
in one call -- what the reference runs as the double loop of blueice/inference.py:392-443 around blueice/model.py:69-91 --
against the oracle (1e-10), against P single-point calls (the same sums grouped by other tiles: 1e-13), with points in one grid
cell and in several, rejected points, two- and four-byte list entries, dataset ranges, results left in HBM, and the
point-by-point route for data the multi-point kernels do not cover."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def _model_and_toys(seed, T=96, shape=(3, 3), bins=(48, 32, 24), counts_max=5):
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel(3, shape, bins, seed=seed)                         # 36 864 bins = 9 tiles of 4096
    ctx = DeviceContext(0)
    m.upload(ctx)
    rng = np.random.default_rng(seed + 1)
    counts = np.zeros((T, m.B))
    for t in range(T):
        hit = rng.choice(m.B, size=int(rng.integers(1500, 4000)), replace=False)
        counts[t, hit] = rng.integers(1, counts_max + 1, size=len(hit))
    counts[5] = 0                                                          # an empty dataset
    ctx.set_param('sparse', 1)
    ctx.upload_counts(counts)
    return m, ctx, counts


def _singles(ctx, z, r, t0=0, t1=None):
    return np.stack([ctx.eval_datasets(z[p], r[p], t0, t1)[0] for p in range(len(z))])


@pytest.mark.parametrize('P', [2, 3, 4, 5, 9, 16])
def test_points_in_several_cells_equal_single_calls_and_the_oracle(P):
    from oracle import blueice_oracle as orc
    m, ctx, counts = _model_and_toys(11)
    try:
        z, r = m.random_points(P, seed=40 + P)
        ll, st = ctx.eval_datasets_points(z, r)
        assert ll.shape == (P, len(counts)) and not st.any() and np.all(np.isfinite(ll))
        assert ctx.get_param('n_toy_points_passes') >= 1                  # the multi-point kernels took the call
        np.testing.assert_allclose(ll, _singles(ctx, z, r), rtol=1e-13, atol=0)
        dense = m.dense_model()
        for p in range(P):
            for t in (0, 5, 50, len(counts) - 1):
                want = orc.loglikelihood(dense, counts[t], z[p], r[p])
                assert abs(ll[p, t] - want) <= RTOL * max(1.0, abs(want)), (p, t, ll[p, t], want)
        again, _ = ctx.eval_datasets_points(z, r)
        np.testing.assert_array_equal(again, ll)                          # fixed summation order: the same bits
    finally:
        ctx.close()


def test_hypotheses_that_differ_in_their_rates_share_the_template_pass():
    """The usual toy-MC scan: one shape point, many signal strengths -- all points in ONE grid cell, so a pass of four points
    is one work item of the log mu kernel (one pass over the cell's rows)."""
    from oracle import blueice_oracle as orc
    m, ctx, counts = _model_and_toys(12)
    try:
        z0, r0 = m.random_points(1, seed=3)
        P = 7
        z = np.repeat(z0, P, axis=0)
        r = np.repeat(r0, P, axis=0)
        r[:, 0] *= np.linspace(0.0, 3.0, P)                               # (a rate of exactly 0 included)
        ll, st = ctx.eval_datasets_points(z, r)
        assert not st.any()
        np.testing.assert_allclose(ll, _singles(ctx, z, r), rtol=1e-13, atol=0)
        dense = m.dense_model()
        for p in (0, 3, P - 1):
            want = orc.loglikelihood(dense, counts[17], z[p], r[p])
            assert abs(ll[p, 17] - want) <= RTOL * abs(want)
        for pp in (2, 4):                                                 # two points per pass: the same values
            ctx.set_param('toy_points_pp', pp)
            got, _ = ctx.eval_datasets_points(z, r)
            np.testing.assert_allclose(got, ll, rtol=1e-13, atol=0)
        ctx.set_param('toy_points_pp', 1)                                 # point by point: bi_eval_datasets' own bits
        before = ctx.get_param('n_toy_points_passes')
        one, _ = ctx.eval_datasets_points(z, r)
        assert ctx.get_param('n_toy_points_passes') == before
        np.testing.assert_array_equal(one, _singles(ctx, z, r))
        ctx.set_param('toy_points_pp', 0)
        for lanes in (2, 4, 8):
            ctx.set_param('toy_points_lanes', lanes)
            got, _ = ctx.eval_datasets_points(z, r)
            np.testing.assert_allclose(got, ll, rtol=1e-13, atol=0)
        ctx.set_param('toy_points_lanes', 0)
    finally:
        ctx.close()


def test_rejected_points_ranges_and_device_output():
    m, ctx, counts = _model_and_toys(13)
    try:
        T = len(counts)
        z, r = m.random_points(6, seed=8)
        z[1, 0] = 99.0                                                    # outside the anchor box   -> -inf, status 1
        r[4, 1] = -1.0                                                    # unphysical rate           -> -inf, status 2
        z[2] = np.nan                                                     # nan z counts as outside   -> -inf, status 1
        ll, st = ctx.eval_datasets_points(z, r)
        assert st.tolist() == [0, 1, 1, 0, 2, 0]
        assert np.isneginf(ll[[1, 2, 4]]).all() and np.all(np.isfinite(ll[[0, 3, 5]]))
        ok = [0, 3, 5]
        np.testing.assert_allclose(ll[ok], _singles(ctx, z[ok], r[ok]), rtol=1e-13, atol=0)
        part, st2 = ctx.eval_datasets_points(z, r, 10, 10 + 70)
        np.testing.assert_array_equal(part, ll[:, 10:80])
        buf = ctx.device_alloc(8 * 6 * T)
        st3 = ctx.eval_datasets_points_device(buf.ptr, z, r)
        np.testing.assert_array_equal(st3, st)
        np.testing.assert_array_equal(buf.to_host(np.float64, 6 * T).reshape(6, T), ll)
        buf.free()
        # every point rejected: nothing to launch but the fill
        allbad, st4 = ctx.eval_datasets_points(z[[1, 2]], r[[1, 2]])
        assert np.isneginf(allbad).all() and st4.tolist() == [1, 1]
    finally:
        ctx.close()


def test_entry_widths_and_the_point_by_point_route():
    """Counts of at most 7: two-byte list entries; larger counts: four-byte entries; a count that fits neither, a non-integer
    or a nan count: the call is answered point by point by bi_eval_datasets' row kernel, with scipy's values."""
    from oracle import blueice_oracle as orc
    m, ctx, counts = _model_and_toys(14)
    try:
        z, r = m.random_points(5, seed=21)
        ll2, _ = ctx.eval_datasets_points(z, r)
        assert ctx.get_param('tmm_entry_bytes') == 2
        big = counts.copy()
        big[3, np.flatnonzero(big[3])[:40]] = 30.0
        ctx.upload_counts(big)
        ll4, _ = ctx.eval_datasets_points(z, r)
        assert ctx.get_param('tmm_entry_bytes') == 4
        keep = np.ones(len(counts), bool)
        keep[3] = False
        np.testing.assert_allclose(ll4[:, keep], ll2[:, keep], rtol=1e-13, atol=0)
        dense = m.dense_model()
        want = orc.loglikelihood(dense, big[3], z[2], r[2])
        assert abs(ll4[2, 3] - want) <= RTOL * abs(want)
        ctx.set_param('dot_entry16', 0)                                   # four-byte entries of the small counts
        ctx.upload_counts(counts)
        wide, _ = ctx.eval_datasets_points(z, r)
        assert ctx.get_param('tmm_entry_bytes') == 4
        np.testing.assert_allclose(wide, ll2, rtol=1e-13, atol=0)
        ctx.set_param('dot_entry16', 1)
        odd = counts.copy()
        odd[8, 100] = 2.5                                                 # scipy: -inf
        odd[9, 7] = np.nan                                                # scipy: nan
        ctx.upload_counts(odd)
        before = ctx.get_param('n_toy_points_passes')
        got, st = ctx.eval_datasets_points(z, r)
        assert ctx.get_param('n_toy_points_passes') == before and not st.any()
        assert np.isneginf(got[:, 8]).all() and np.isnan(got[:, 9]).all()
        keep = np.ones(len(counts), bool)
        keep[[8, 9]] = False
        np.testing.assert_allclose(got[:, keep], ll2[:, keep], rtol=1e-13, atol=0)
    finally:
        ctx.close()


@pytest.mark.parametrize('seed', range(4))
def test_random_shapes(seed):
    """Random bin counts (last tile partial), dataset counts, fill levels, d = 0 ... 2 shape parameters, points per call."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    rng = np.random.default_rng(900 + seed)
    d = int(rng.integers(0, 3))
    bins = (int(rng.integers(130, 200)), int(rng.integers(100, 140)))     # 13 000 ... 28 000 bins: 4 ... 7 tiles
    m = SyntheticModel(int(rng.integers(1, 5)), (3,) * d, bins, seed=seed)
    ctx = DeviceContext(0)
    try:
        m.upload(ctx)
        T = int(rng.integers(64, 200))
        counts = np.zeros((T, m.B))
        n_tl = -(-m.B // 4096)
        for t in range(T):                                                # 10 ... 90 entries per (dataset, tile) run: runs that fit the
            hit = rng.choice(m.B, size=int(rng.integers(10, 90)) * n_tl, replace=False)   # kernel's slots, and runs with a tail
            counts[t, hit] = rng.integers(1, 12 if seed % 2 else 7, size=len(hit))
        ctx.set_param('sparse', 1)
        ctx.upload_counts(counts)
        P = int(rng.integers(2, 12))
        z, r = m.random_points(P, seed=seed)
        ll, st = ctx.eval_datasets_points(z if d else None, r)
        assert not st.any() and ll.shape == (P, T)
        single = np.stack([ctx.eval_datasets(z[p] if d else None, r[p])[0] for p in range(P)])
        np.testing.assert_allclose(ll, single, rtol=1e-13, atol=0)
    finally:
        ctx.close()


def test_second_stream_gives_the_same_bits():
    """toy_points_overlap = 1: the next group's log mu pass and the last group's finish on a second stream beside the dot kernel, on
    their own halves of the scratch buffers -- 20 points in random cells (three groups of passes), twice, bitwise what one stream gives."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel(3, (3, 3), (150, 120), seed=5)
    ctx = DeviceContext(0)
    try:
        m.upload(ctx)
        rng = np.random.default_rng(77)
        T = 150
        counts = np.zeros((T, m.B))
        for t in range(T):
            hit = rng.choice(m.B, size=300, replace=False)
            counts[t, hit] = rng.integers(1, 7, size=300)
        ctx.set_param('sparse', 1)
        ctx.upload_counts(counts)
        z, r = m.random_points(20, seed=9)
        z[7, 0] = 99.0                                                    # a rejected point among them
        one, st1 = ctx.eval_datasets_points(z, r)
        before = ctx.get_param('n_toy_points_passes')
        ctx.set_param('toy_points_overlap', 1)
        for _ in range(2):
            two, st2 = ctx.eval_datasets_points(z, r)
            np.testing.assert_array_equal(st1, st2)
            np.testing.assert_array_equal(one, two)
        assert ctx.get_param('n_toy_points_passes') == before + 2 * 5     # 19 valid points: five passes per call
        part, _ = ctx.eval_datasets_points(z, r, 10, 140)
        np.testing.assert_array_equal(part, one[:, 10:140])
    finally:
        ctx.close()


def test_through_the_likelihood_class():
    """`lf.eval_toys_points(points)` -- the reference-style entry: a dict of parameter arrays in, ll [P, T] out -- equals one
    `lf.eval_toys(**kw)` per hypothesis (priors, defaults and rejected points included)."""
    import model_zoo
    ns = model_zoo.namespace_of('blueice_amd')
    lf, _, _ = model_zoo.d3_small(ns)
    lf.simulate_toys(200, seed=5, shift=0.2, stretch=-0.3, tilt=0.5, s1_rate_multiplier=2.0)
    P = 9
    rng = np.random.default_rng(3)
    points = dict(shift=rng.uniform(-0.9, 0.9, P), stretch=rng.uniform(-0.9, 0.9, P), tilt=np.full(P, 0.5),
                  s1_rate_multiplier=np.linspace(0.0, 3.0, P), s0_rate_multiplier=1.3)
    points['shift'][4] = 99.0                                            # outside the anchor box: a row of -inf
    got = lf.eval_toys_points(points)
    assert got.shape == (P, 200) and np.isneginf(got[4]).all()
    for i in range(P):
        kw = {k: float(np.broadcast_to(v, (P,))[i]) for k, v in points.items()}
        want = lf.eval_toys(**kw)
        if i == 4:
            assert np.isneginf(want).all()
        else:
            np.testing.assert_allclose(got[i], want, rtol=1e-13, atol=0)
    part = lf.eval_toys_points(points, t0=20, t1=120)
    np.testing.assert_array_equal(part, got[:, 20:120])
