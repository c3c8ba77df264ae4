"""Randomised small configurations against the CPU oracle: every combination of shape-axis count (including
single-anchor axes), source count, ragged bin counts around the 512-bin tile, zero templates, empty bins,
on-anchor / corner points, batch grouping (1..16 points per cell pass), both data forms, gradient, toys."""
import itertools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10
# BLUEICE_FUZZ_SEEDS=first:count widens the campaign (default: the 12 + 8 seeds below)
_FIRST, _COUNT = (int(v) for v in os.environ.get('BLUEICE_FUZZ_SEEDS', '0:0').split(':'))


def random_case(rng, d, S, B, bb):
    n_anchor = [int(rng.integers(1, 5)) for _ in range(d)]
    if d and all(n == 1 for n in n_anchor):
        n_anchor[0] = 2
    anchor_z = [np.sort(rng.uniform(-3, 3, n)) if n > 1 else np.array([rng.uniform(-1, 1)]) for n in n_anchor]
    shape = tuple(n_anchor)
    ps = rng.random(shape + (S, B)) ** 3
    ps[rng.random(ps.shape) < 0.15] = 0.0                      # exact zeros: mu = 0 bins happen
    ps /= np.maximum(ps.sum(axis=-1, keepdims=True), 1e-300)
    mus = rng.uniform(5, 60, shape + (S,))
    n_model = None
    if bb >= 0:
        n_model = np.ones(shape + (S, B))
        n_model[..., bb, :] = 1.0 + rng.poisson(20., shape + (B,))
        ps = np.maximum(ps, 1e-6)
    lam = (mus.reshape(-1, S)[0][:, None] * ps.reshape(-1, S, B)[0]).sum(axis=0)
    counts = rng.poisson(lam * rng.choice([0.3, 1.0, 3.0])).astype(float)
    return dict(anchor_z=anchor_z, ps=ps, mus=mus, n_model=n_model), counts


def random_points(rng, model, P, S):
    zs = []
    for _ in range(P):
        z = []
        for g in model['anchor_z']:
            mode = rng.integers(0, 5)
            if len(g) == 1 or mode == 0:
                z.append(float(rng.choice(g)))                  # exactly on an anchor
            elif mode == 1:
                z.append(float(g[-1]))                          # top edge
            else:
                z.append(float(rng.uniform(g[0], g[-1])))
        zs.append(z)
    r = rng.uniform(0.2, 2.0, size=(P, S))
    r[rng.random((P, S)) < 0.1] = 0.0                           # switched-off sources
    return np.array(zs, dtype=float).reshape(P, len(model['anchor_z'])), r


CONFIGS = list(itertools.product([0, 1, 2, 3, 4], [1, 3, 7], [1, 37, 511, 512, 513, 1300]))


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(12))
def test_random_configurations_match_oracle(seed):
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(1000 + seed)
    ctx = DeviceContext(0)
    for _ in range(6):
        d, S, B = CONFIGS[int(rng.integers(len(CONFIGS)))]
        bb = int(rng.integers(S)) if rng.random() < 0.3 else -1
        model, counts = random_case(rng, d, S, B, bb)
        P = int(rng.integers(1, 40))
        z, r = random_points(rng, model, P, S)
        if d and rng.random() < 0.5:                            # make several points share a cell
            z[P // 2:] = z[0] + 0 * z[P // 2:]
            z[P // 2:, 0] = np.clip(z[0, 0] + rng.uniform(-1e-3, 1e-3, P - P // 2), model['anchor_z'][0][0],
                                    model['anchor_z'][0][-1])
        want = orc.loglikelihood_batch(model, counts, z, r, bb_source=bb if bb >= 0 else None)
        for sparse, maxg in ((0, 16), (2, 4), (2, 1)):
            ctx.set_param('sparse', sparse)
            ctx.set_param('max_group', maxg)
            ctx.upload_model(model['anchor_z'], model['ps'], model['mus'], n_model=model['n_model'], bb_source=bb)
            ctx.upload_counts(counts)
            got, st = ctx.eval(z if d else None, r)
            singles = [ctx.eval(z[i] if d else None, r[i]) for i in range(min(P, 5))]
            one = np.array([a[0] for a, _ in singles])
            one_st = np.array([b[0] for _, b in singles])
            for i in range(P):
                if bb >= 0:
                    # the reference asserts (want is nan) exactly where the device raises its assertion bits -- the knife
                    # edge at U_b == 0 included, in batched passes as in single calls (N in numpy's summation order)
                    assert bool(st[i] & 12) == bool(np.isnan(want[i])), (seed, d, S, B, bb, sparse, maxg, i, int(st[i]), want[i])
                    if np.isnan(want[i]):
                        continue
                ok = (got[i] == want[i]) if not np.isfinite(want[i]) else abs(got[i] - want[i]) <= RTOL * max(1, abs(want[i]))
                assert ok, (seed, d, S, B, bb, sparse, maxg, i, got[i], want[i])
            for i in range(len(one)):
                if np.isfinite(want[i]):
                    assert abs(one[i] - want[i]) <= RTOL * max(1, abs(want[i]))
                if bb >= 0:
                    # single-point calls see the reference's own bits (N in numpy's summation order): the assertion
                    # status must agree EXACTLY, knife edge at U_b == 0 included -- no forgiveness here
                    assert bool(one_st[i] & 12) == bool(np.isnan(want[i])), (seed, d, S, B, bb, sparse, i, int(one_st[i]), want[i])
            if bb < 0:
                ll, gz, gs, _ = ctx.eval_grad(z if d else None, r)
                fin = np.isfinite(want)
                np.testing.assert_allclose(ll[fin], want[fin], rtol=1e-10, atol=1e-10)
                toys, _ = ctx.eval_datasets(z[0] if d else None, r[0])
                if np.isfinite(want[0]):
                    assert abs(toys[0] - want[0]) <= RTOL * max(1, abs(want[0]))
                else:
                    assert toys[0] == want[0] or (np.isnan(toys[0]) and np.isnan(want[0]))
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(8))
def test_large_batches_device_planner_and_scan_kernel(seed):
    """Batches large enough for the device-side planner and the matrix-core scan kernel, on random small models:
    several datasets, counts with nan / negative / non-integer entries, sources allowed to go negative (mu < 0 ->
    nan), switched-off sources, points outside the box, nan coordinates, unphysical rates, bad dataset numbers."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(5000 + seed)
    ctx = DeviceContext(0)
    scan_runs = 0
    for rep in range(3):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([1, 2, 3, 4, 5, 8]))
        B = int(rng.choice([40, 511, 512, 700, 1300]))
        model, counts0 = random_case(rng, d, S, B, -1)
        T = int(rng.integers(1, 4))
        counts = np.stack([rng.poisson(counts0 * rng.uniform(0.5, 2)).astype(float) for _ in range(T)])
        if rng.random() < 0.5:
            counts[rng.integers(T), rng.integers(B)] = rng.choice([np.nan, -1.0, 2.5])
        allow_negative = None
        if rng.random() < 0.4:
            allow_negative = np.zeros(S, dtype=bool)
            allow_negative[rng.integers(S)] = True
        P = int(rng.integers(600, 1500))
        z, r = random_points(rng, model, P, S)
        if allow_negative is not None:
            neg = rng.random(P) < 0.2
            r[neg, np.flatnonzero(allow_negative)[0]] = -rng.uniform(0.5, 3.0, neg.sum())
        ds = rng.integers(0, T, P)
        bad = rng.random(P)
        if d:
            z[bad < 0.02, 0] = 99.0
            z[(bad > 0.02) & (bad < 0.03), d - 1] = np.nan
        r[(bad > 0.03) & (bad < 0.05), 0] = -0.5 if allow_negative is None or not allow_negative[0] else np.inf
        ds[(bad > 0.05) & (bad < 0.06)] = T + 3
        want = np.empty(P)
        for i in range(P):
            if not 0 <= ds[i] < T:
                want[i] = np.nan
                continue
            want[i] = orc.loglikelihood(model, counts[ds[i]], z[i], r[i], allow_negative=allow_negative)
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
        if allow_negative is not None:
            ctx.set_allow_negative(allow_negative)
        for sparse in (0, 1):
            ctx.set_param('sparse', sparse)
            ctx.upload_counts(counts)
            before = ctx.get_param('n_scan_launches') + ctx.get_param('n_valid_launches')
            got, st = ctx.eval(z if d else None, r, dataset=ds)
            used_scan = ctx.get_param('n_scan_launches') + ctx.get_param('n_valid_launches') > before
            scan_runs += used_scan
            bad_ds = ~((ds >= 0) & (ds < T))
            assert np.all((st[bad_ds] & 16) != 0) and not np.any(st[~bad_ds] & 16)
            for i in np.flatnonzero(~bad_ds):
                w, g = want[i], got[i]
                ok = (np.isnan(w) and np.isnan(g)) or g == w or (np.isfinite(w) and abs(g - w) <= RTOL * max(1, abs(w)))
                # (infinite rates -- legal only next to sources that may go negative, likelihood.py:403-415 -- are
                # answered the reference's way on the host, inf_rate_value: no escape clause here any more)
                assert ok, (seed, rep, d, S, B, T, sparse, used_scan, i, g, w, st[i])
        if allow_negative is not None:
            ctx.set_allow_negative(np.zeros(S, dtype=bool))
    print('matrix-core scan kernel used in %d of 6 batches' % scan_runs)
    assert scan_runs >= 1 or _COUNT      # (with the default seeds every case reaches the scan kernel at least once)
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_unbinned_random_configurations_match_oracle(seed):
    """The extended unbinned likelihood (rows = pdf values at the events, likelihood.py:531-573,678-690) on random
    models: events where every source's pdf is 0 (the outlier clamp), no events at all, small and large batches."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(9000 + seed)
    ctx = DeviceContext(0)
    for rep in range(4):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([1, 2, 3, 5]))
        n_ev = int(rng.choice([0, 1, 7, 300, 513, 2000]))
        model, _ = random_case(rng, d, S, max(n_ev, 1), -1)
        shape = model['ps'].shape[:-2]
        ps = rng.random(shape + (S, n_ev)) ** 2
        ps[rng.random(ps.shape) < 0.2] = 0.0
        if n_ev:
            ps[..., :, rng.integers(n_ev)] = 0.0                 # an event no source can explain, at every anchor
        model = dict(anchor_z=model['anchor_z'], ps=ps, mus=model['mus'], n_model=None)
        outlier = float(rng.choice([1e-12, 1e-7, 0.0]))
        ctx.begin_model(model['anchor_z'], S, n_ev)
        n_anchors = int(np.prod(shape, dtype=np.int64))
        flat_ps, flat_mus = ps.reshape((n_anchors, S, n_ev)), model['mus'].reshape((n_anchors, S))
        for a in range(len(flat_mus)):
            ctx.set_anchor(a, flat_ps[a], flat_mus[a])
        ctx.end_model()
        ctx.set_unbinned(outlier)
        for P in (int(rng.integers(1, 30)), int(rng.integers(600, 900))):
            z, r = random_points(rng, model, P, S)
            with np.errstate(all='ignore'):
                want = np.array([orc.loglikelihood_unbinned(model, z[i], r[i], outlier_likelihood=outlier) for i in range(P)])
            got, st = ctx.eval(z if d else None, r)
            one, _ = ctx.eval(z[0] if d else None, r[0])
            for i in range(P):
                w, g = want[i], got[i]
                ok = (np.isnan(w) and np.isnan(g)) or g == w or (np.isfinite(w) and abs(g - w) <= RTOL * max(1, abs(w)))
                assert ok, (seed, rep, d, S, n_ev, outlier, P, i, g, w, st[i])
            assert (np.isnan(want[0]) and np.isnan(one[0])) or one[0] == want[0] or abs(one[0] - want[0]) <= RTOL * max(1, abs(want[0]))
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_gradient_random_configurations_against_finite_differences(seed):
    """bi_eval_grad on random models, both data forms: d ll / d z and d ll / d rate_scale against central finite
    differences of the CPU oracle at points strictly inside grid cells (ll has kinks on the anchors)."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(7000 + seed)
    ctx = DeviceContext(0)
    for rep in range(4):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([1, 2, 3, 5]))
        B = int(rng.choice([1, 37, 512, 700]))
        model, counts = random_case(rng, d, S, B, -1)
        model['ps'] = np.maximum(model['ps'], 1e-4)              # keep mu away from 0: ll is smooth in the cell
        model['ps'] /= model['ps'].sum(axis=-1, keepdims=True)
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
        P = 6
        z = np.empty((P, d))
        h = np.empty(d)
        for ax, g in enumerate(model['anchor_z']):
            if len(g) == 1:
                z[:, ax], h[ax] = g[0], 0.0                      # a single-anchor axis has nothing to differentiate
                continue
            k = rng.integers(0, len(g) - 1, P)
            z[:, ax] = g[k] + (g[k + 1] - g[k]) * rng.uniform(0.2, 0.8, P)
            h[ax] = 1e-5 * np.min(np.diff(g))
        r = rng.uniform(0.3, 2.0, (P, S))
        for sparse in (0, 1):
            ctx.set_param('sparse', sparse)
            ctx.upload_counts(counts)
            ll, gz, gs, st = ctx.eval_grad(z if d else None, r)
            assert not st.any()
            for i in range(P):
                f = lambda zz, rr: orc.loglikelihood(model, counts, zz, rr)
                want = f(z[i], r[i])
                assert abs(ll[i] - want) <= RTOL * max(1, abs(want))
                scale = max(1.0, abs(want))
                for ax in range(d):
                    if h[ax] == 0.0:
                        continue
                    e = np.zeros(d); e[ax] = h[ax]
                    fd = (f(z[i] + e, r[i]) - f(z[i] - e, r[i])) / (2 * h[ax])
                    assert abs(gz[i, ax] - fd) <= 1e-4 * max(abs(fd), scale / max(np.ptp(model['anchor_z'][ax]), 1e-9) * 1e-3), \
                        (seed, rep, d, S, B, sparse, i, ax, gz[i, ax], fd)
                for s in range(S):
                    e = np.zeros(S); e[s] = 1e-6 * r[i, s]
                    fd = (f(z[i], r[i] + e) - f(z[i], r[i] - e)) / (2 * e[s])
                    assert abs(gs[i, s] - fd) <= 1e-4 * max(abs(fd), 1e-3 * scale), (seed, rep, d, S, B, sparse, i, s, gs[i, s], fd)
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_beeston_barlow_gradient_random_configurations(seed):
    """bi_eval_grad with Beeston-Barlow on random models (0-3 shape axes incl. single-anchor ones, 1-5 sources, any source
    as the Beeston-Barlow one): value = bi_eval's, slopes against central differences of the oracle inside grid cells.
    With a single source every bin has U_b == 0 -- the reference's special case, differentiated as such."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(7600 + seed)
    ctx = DeviceContext(0)
    for rep in range(4):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([1, 2, 3, 5]))
        B = int(rng.choice([3, 37, 512, 700]))
        bb = int(rng.integers(0, S))
        model, counts = random_case(rng, d, S, B, bb)
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'], n_model=model['n_model'], bb_source=bb)
        ctx.upload_counts(counts)
        P = 5
        z = np.empty((P, d))
        h = np.empty(d)
        for ax, g in enumerate(model['anchor_z']):
            if len(g) == 1:
                z[:, ax], h[ax] = g[0], 0.0
                continue
            k = rng.integers(0, len(g) - 1, P)
            z[:, ax] = g[k] + (g[k + 1] - g[k]) * rng.uniform(0.2, 0.8, P)
            h[ax] = 1e-5 * np.min(np.diff(g))
        r = rng.uniform(0.3, 2.0, (P, S))
        ll, gz, gs, st = ctx.eval_grad(z if d else None, r)
        ref, rst = ctx.eval(z if d else None, r)
        np.testing.assert_array_equal(st, rst)
        f = lambda zz, rr: orc.loglikelihood(model, counts, zz, rr, bb_source=bb, forgive_zero_u=True)
        for i in np.flatnonzero(st == 0):
            want = f(z[i], r[i])
            assert abs(ll[i] - ref[i]) <= 1e-13 * max(1, abs(ref[i])) and abs(ll[i] - want) <= RTOL * max(1, abs(want))
            scale = max(1.0, abs(want))
            for ax in range(d):
                if h[ax] == 0.0:
                    continue
                e = np.zeros(d); e[ax] = h[ax]
                fd = (f(z[i] + e, r[i]) - f(z[i] - e, r[i])) / (2 * h[ax])
                assert abs(gz[i, ax] - fd) <= 1e-4 * max(abs(fd), scale / max(np.ptp(model['anchor_z'][ax]), 1e-9) * 1e-3), \
                    (seed, rep, d, S, B, bb, i, ax, gz[i, ax], fd)
            for s in range(S):
                e = np.zeros(S); e[s] = 1e-6 * r[i, s]
                fd = (f(z[i], r[i] + e) - f(z[i], r[i] - e)) / (2 * e[s])
                assert abs(gs[i, s] - fd) <= 1e-4 * max(abs(fd), 1e-3 * scale), (seed, rep, d, S, B, bb, i, s, gs[i, s], fd)
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_device_histogram_random_spaces_equal_numpy_histogramdd(seed):
    """bi_upload_events (set_data on the device, likelihood.py:603-609) on random analysis spaces: 1-4 axes, uniform and
    non-uniform edges, events exactly on edges (interior, first, last), outside, +-inf, nan."""
    from blueice_amd.device import DeviceContext
    rng = np.random.default_rng(11000 + seed)
    ctx = DeviceContext(0)
    for rep in range(5):
        k = int(rng.integers(1, 5))
        edges = []
        for _ in range(k):
            n = int(rng.integers(1, 9))
            e = np.linspace(rng.uniform(-5, 0), rng.uniform(1, 6), n + 1) if rng.random() < 0.5 \
                else np.cumsum(rng.uniform(0.1, 2.0, n + 1)) - rng.uniform(0, 5)
            edges.append(e)
        shape = tuple(len(e) - 1 for e in edges)
        B = int(np.prod(shape))
        ctx.upload_model([], np.full((1, B), 1.0 / B), np.array([10.]))
        ctx.set_analysis_space(edges)
        N = int(rng.choice([0, 1, 50, 5000]))
        cols = []
        for e in edges:
            x = rng.uniform(e[0] - 1, e[-1] + 1, N)
            snap = rng.random(N) < 0.3
            x[snap] = rng.choice(e, snap.sum())                  # exactly on an edge
            odd = rng.random(N) < 0.02
            x[odd] = rng.choice([np.inf, -np.inf, np.nan], odd.sum())
            cols.append(x)
        ctx.upload_events(*cols)
        got = ctx.download_counts(0).reshape(shape)
        sample = np.stack(cols, axis=1) if N else np.zeros((0, k))
        keep = ~np.isnan(sample).any(axis=1)                     # numpy refuses nan in the range autodetection only; drop them
        want = np.histogramdd(sample[keep], bins=edges)[0]
        np.testing.assert_array_equal(got, want, err_msg=str((seed, rep, shape, N)))
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_toy_evaluation_random_configurations_match_oracle(seed):
    """bi_eval_datasets (one point, T datasets: the toy-MC call shape) on random models, both data forms (dense
    counts / non-empty-bin lists), dataset sub-ranges, datasets with invalid counts, rejected points."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(13000 + seed)
    ctx = DeviceContext(0)
    for rep in range(4):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([1, 2, 4, 7]))
        B = int(rng.choice([1, 40, 512, 1300]))
        model, counts0 = random_case(rng, d, S, B, -1)
        T = int(rng.choice([1, 3, 17, 40]))
        counts = np.stack([rng.poisson(counts0 * rng.uniform(0.3, 3)).astype(float) for _ in range(T)])
        if rng.random() < 0.5:
            counts[rng.integers(T), rng.integers(B)] = rng.choice([np.nan, -1.0, 2.5])
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
        z, r = random_points(rng, model, 3, S)
        if d:
            z[2, 0] = 77.0                                       # outside the box
        for sparse in (0, 1):
            ctx.set_param('sparse', sparse)
            ctx.upload_counts(counts)
            for i in range(3):
                t0 = int(rng.integers(0, T))
                t1 = int(rng.integers(t0, T + 1))
                got, st = ctx.eval_datasets(z[i] if d else None, r[i], t0, t1)
                with np.errstate(all='ignore'):
                    want = np.array([orc.loglikelihood(model, counts[t], z[i], r[i]) for t in range(t0, t1)])
                assert got.shape == want.shape
                for t in range(len(want)):
                    w, g = want[t], got[t]
                    ok = (np.isnan(w) and np.isnan(g)) or g == w or (np.isfinite(w) and abs(g - w) <= RTOL * max(1, abs(w)))
                    assert ok, (seed, rep, d, S, B, T, sparse, i, t0 + t, g, w, st)
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_toy_points_random_configurations_match_oracle(seed):
    """bi_eval_datasets_points (P points x T datasets: the toy-MC call over several hypotheses) on random models against the
    oracle: hypotheses in one cell and in several, rejected points among them, dataset sub-ranges, datasets with invalid counts,
    both list entry widths, passes of 2 and of 4 points, every lane split of the dot kernel."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(17000 + seed)
    ctx = DeviceContext(0)
    multi_runs = 0
    for rep in range(3):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([1, 2, 4, 7]))
        big = rep != 1                     # large enough for the multi-point kernels (four bin tiles, 64 datasets); else the point-by-point route
        B = int(rng.choice([12289, 16384, 20001]) if big else rng.choice([1, 40, 1300, 4097, 9000]))
        model, counts0 = random_case(rng, d, S, B, -1)
        T = int(rng.choice([64, 97, 130]) if big else rng.choice([1, 3, 17, 70]))
        if big:      # 12 ... 80 events per dataset and bin tile, drawn from the first anchor's expectation
            lam = (model['mus'].reshape(-1, S)[0][:, None] * model['ps'].reshape(-1, S, B)[0]).sum(axis=0)
            lam *= rng.uniform(12, 80) * -(-B // 4096) / lam.sum()
            counts = rng.poisson(lam, size=(T, B)).astype(float)
        else:
            counts = np.stack([rng.poisson(counts0 * rng.uniform(0.3, 3)).astype(float) for _ in range(T)])
        invalid_count = rng.random() < 0.3
        if invalid_count:
            counts[rng.integers(T), rng.integers(B)] = rng.choice([np.nan, -1.0, 2.5])
        if rng.random() < 0.3:                                          # beyond a two-byte entry: four-byte lists; beyond those
            huge = float(rng.choice([30000.0, 30000.0, 70000.0]))       # (32 767): point by point
            counts[rng.integers(T), rng.integers(B)] = huge
            invalid_count |= huge > 32767
        if rng.random() < 0.2:
            counts[rng.integers(T)] = 0.0                               # a dataset without events
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
        P = int(rng.choice([2, 3, 5, 8, 9]) if big else rng.choice([1, 2, 5, 13]))
        z, r = random_points(rng, model, P, S)
        if d and P > 2 and rng.random() < 0.6:
            z[1:P // 2 + 1] = z[0]                                       # several hypotheses of one cell, apart in their rates
        if d and rng.random() < 0.5:
            z[rng.integers(P), 0] = 77.0                                 # outside the box
        if rng.random() < 0.3:
            r[rng.integers(P), 0] = -1.0                                 # unphysical
        ctx.set_param('sparse', 1)
        ctx.set_param('dot_entry16', int(rng.random() < 0.8))
        ctx.upload_counts(counts)
        ctx.set_param('toy_points_pp', int(rng.choice([0, 0, 2, 4])))
        ctx.set_param('toy_points_lanes', int(rng.choice([0, 0, 2, 4, 8])))
        for i in range(2):
            t0 = int(rng.integers(0, T))
            t1 = int(rng.integers(t0, T + 1)) if i else T
            t0 = t0 if i else 0
            before = ctx.get_param('n_toy_points_passes')
            got, st = ctx.eval_datasets_points(z if d else None, r, t0, t1)
            ran_multi = ctx.get_param('n_toy_points_passes') > before
            multi_runs += ran_multi
            # (lists with an invalid count go point by point, like ranges of fewer than 64 datasets; rejected points are rows of -inf)
            if big and not invalid_count and i == 0 and (st == 0).any():
                assert ran_multi, repr((seed, rep, d, S, B, T, P, st.tolist(), int((counts > 0).sum()), float(np.nanmax(counts)), {k: ctx.get_param(k) for k in ('tmm_entry_bytes', 'toy_points_pp', 'toy_points_lanes', 'dot_entry16', 'sparse')}))
            assert got.shape == (P, t1 - t0)
            with np.errstate(all='ignore'):
                for p in range(P):
                    for t in range(t0, t1):
                        w, g = orc.loglikelihood(model, counts[t], z[p], r[p]), got[p, t - t0]
                        ok = (np.isnan(w) and np.isnan(g)) or g == w or (np.isfinite(w) and abs(g - w) <= RTOL * max(1, abs(w)))
                        assert ok, (seed, rep, d, S, B, T, P, p, t, g, w, st)
        ctx.set_param('toy_points_pp', 0)
        ctx.set_param('toy_points_lanes', 0)
        ctx.set_param('dot_entry16', 1)
    ctx.close()
    print('multi-point kernels used in %d of 6 calls' % multi_runs)


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_full_output_random_configurations_match_oracle(seed):
    """bi_eval_full / bi_interpolate (full_output=True and the morpher closures, likelihood.py:355-357,424-425) on random
    models with and without Beeston-Barlow: the interpolated tensors bit for bit, the adjusted (mus, ps) and ll to 1e-10."""
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    rng = np.random.default_rng(15000 + seed)
    ctx = DeviceContext(0)
    for rep in range(4):
        d = int(rng.integers(0, 4))
        S = int(rng.choice([2, 3, 5]))
        B = int(rng.choice([1, 37, 512, 700]))
        bb = int(rng.integers(S)) if rng.random() < 0.5 else -1
        model, counts = random_case(rng, d, S, B, bb)
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'], n_model=model['n_model'], bb_source=bb)
        ctx.upload_counts(counts)
        z, r = random_points(rng, model, 4, S)
        r = np.maximum(r, 0.2)                                    # every source on: away from the U == 0 knife edge
        for i in range(4):
            zi = z[i] if d else None
            ps_want = orc.interpolate(model['anchor_z'], model['ps'], z[i])
            np.testing.assert_array_equal(ctx.interpolate('ps', zi), ps_want.reshape(S, B))
            mus_want = orc.rates_at(model, z[i], np.ones(S))
            np.testing.assert_array_equal(ctx.interpolate('mus', zi), mus_want)
            ll, mus, ps, st = ctx.eval_full(zi, r[i])
            mus_o, ps_o = mus_want * r[i], ps_want.reshape(S, B)
            if bb >= 0:
                nm = orc.interpolate(model['anchor_z'], model['n_model'], z[i]).reshape(S, B)
                np.testing.assert_array_equal(ctx.interpolate('n_model', zi), nm[bb])
                try:
                    mus_o, ps_o = orc.adjust_expectations_bb(mus_o, ps_o, nm, counts, bb)
                except AssertionError:
                    assert st & 12
                    continue
            want = orc.compute_likelihood(mus_o, ps_o, counts)
            assert st == 0, (seed, rep, d, S, B, bb, i, st)
            np.testing.assert_allclose(mus, mus_o, rtol=1e-10)
            np.testing.assert_allclose(ps, ps_o, rtol=1e-10, atol=1e-300)
            assert (np.isfinite(want) and abs(ll - want) <= RTOL * max(1, abs(want))) or ll == want
    ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(4))
def test_likelihood_api_random_calls_match_oracle_plus_priors(seed):
    """The Python half of a call (likelihood.py:328-415: defaults, rate multipliers, live-time scaling, log priors,
    bounds) on top of the device half: lf(**kw) on a synthetic model with random priors against
    oracle(tensors) + priors, for scalar calls, eval_points and LogLikelihoodSum."""
    from scipy import stats
    from oracle import blueice_oracle as orc
    from blueice_amd import LogLikelihoodSum
    from blueice_amd.likelihood import BinnedLogLikelihood
    from blueice_amd.synthetic import SyntheticModel, TemplateSource
    rng = np.random.default_rng(17000 + seed)
    S = int(rng.integers(1, 5))
    n_anchor = tuple(int(rng.integers(2, 4)) for _ in range(int(rng.integers(1, 4))))
    bins = tuple(int(rng.integers(2, 9)) for _ in range(int(rng.integers(1, 3))))
    m = SyntheticModel(S, n_anchor, bins, seed=int(rng.integers(1 << 30)))
    names = ['shape%d' % i for i in range(m.d)]
    config = dict(analysis_space=[('x%d' % i, np.arange(b + 1, dtype=float)) for i, b in enumerate(bins)],
                  default_source_class=TemplateSource, synthetic_model=m, shape_names=names, livetime_days=2.0,
                  sources=[dict(name='s%d' % s, source_index=s) for s in range(S)],
                  **{n: float(g[len(g) // 2]) for n, g in zip(names, m.anchor_z)})
    lf = BinnedLogLikelihood(config)
    rate_priors, shape_priors = {}, {}
    for s in range(S):
        if rng.random() < 0.5:
            rate_priors[s] = stats.norm(1.0, float(rng.uniform(0.05, 0.5))).logpdf
        lf.add_rate_parameter('s%d' % s, log_prior=rate_priors.get(s))
    for i, (n, g) in enumerate(zip(names, m.anchor_z)):
        if rng.random() < 0.5:
            shape_priors[i] = stats.norm(float(g[len(g) // 2]), float(rng.uniform(0.3, 2.0))).logpdf
        lf.add_shape_parameter(n, tuple(float(v) for v in g), log_prior=shape_priors.get(i))
    lf.prepare()
    counts = m.counts(dense=True)
    lf.set_binned_data(counts.reshape(bins))
    model = m.dense_model()
    # TemplateSource expectations are per day at livetime 1; the config's livetime_days = 2 doubles them
    model = dict(model, mus=model['mus'] * 1.0)

    def expected(z, mult, livetime):
        scale = np.array(mult, dtype=float) * (1.0 if livetime is None else livetime / 2.0)
        ll = orc.loglikelihood(model, counts, z, scale)
        prior = sum(p(mult[s]) for s, p in rate_priors.items()) + sum(p(z[i]) for i, p in shape_priors.items())
        return ll if ll == -np.inf else ll + prior

    base_z = np.array([g[len(g) // 2] for g in m.anchor_z])
    calls = []
    for _ in range(12):
        kw, z, mult = {}, base_z.copy(), np.ones(S)
        for i, (n, g) in enumerate(zip(names, m.anchor_z)):
            if rng.random() < 0.6:
                z[i] = rng.uniform(g[0] - 0.2, g[-1] + 0.2) if rng.random() < 0.9 else rng.choice(g)
                kw[n] = float(z[i])
        for s in range(S):
            if rng.random() < 0.6:
                mult[s] = rng.choice([0.0, rng.uniform(0.1, 3.0)])
                kw['s%d_rate_multiplier' % s] = float(mult[s])
        livetime = float(rng.uniform(0.5, 5.0)) if rng.random() < 0.4 else None
        calls.append((kw, z.copy(), mult.copy(), livetime))
        want = expected(z, mult, livetime)
        got = lf(livetime_days=livetime, **kw)
        assert got == want or abs(got - want) <= RTOL * max(1, abs(want)), (seed, kw, livetime, got, want)
    # the batched form over the same calls (those without a live-time override), and a weighted sum of two copies
    plain = [c for c in calls if c[3] is None]
    if plain:
        pts = {n: np.array([c[1][i] for c in plain]) for i, n in enumerate(names)}
        pts.update({'s%d_rate_multiplier' % s: np.array([c[2][s] for c in plain]) for s in range(S)})
        want = np.array([expected(c[1], c[2], None) for c in plain])
        got = lf.eval_points(pts)
        fin = np.isfinite(want)
        assert np.array_equal(got[~fin], want[~fin])
        np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)
        tot = LogLikelihoodSum([lf, lf], likelihood_weights=[1, 0.5])      # same context twice: the sequential form
        kw = plain[0][0]
        assert tot(**kw) == lf(**kw) + 0.5 * lf(**kw)


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(8))
def test_fuzz_event_scoring(seed):
    """bi_score_events (unbinned set_data on the device) against scipy's RegularGridInterpolator over the bin centres /
    the bin lookup of the package's Histdd: random numbers of axes, sources, anchors, non-uniform edges, events on
    centres, edges and range limits.  The interpolating method is held to the bit wherever scipy takes its general
    routine (not 2 axes); the likelihood built from the scored tensor is compared with the oracle's."""
    from scipy.interpolate import RegularGridInterpolator
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    from blueice_amd.histdd import Histdd
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.integers(1, 6))            # (up to 3 axes: four events per thread, corners unrolled; 4, 5: one event, a corner loop)
    S = int(rng.integers(1, 4))
    d = int(rng.integers(0, 3))
    method = 'linear' if rng.random() < 0.6 else 'piecewise'
    edges = [np.sort(rng.uniform(-3, 3, int(rng.integers(3, 8)))) for _ in range(k)]
    edges = [e + 1e-3 * np.arange(len(e)) for e in edges]                     # strictly ascending
    shape = tuple(len(e) - 1 for e in edges)
    B = int(np.prod(shape))
    anchor_z = [np.sort(rng.uniform(-2, 2, int(rng.integers(2, 4)))) + 1e-3 * np.arange(1) for _ in range(d)]
    anchor_z = [np.unique(g) for g in anchor_z]
    if any(len(g) < 2 for g in anchor_z):
        anchor_z = [np.array([-1., 0.5, 1.25])[:max(2, len(g))] for g in anchor_z]
    A = int(np.prod([len(g) for g in anchor_z])) if d else 1
    dens = rng.random((A, S) + shape) ** 2
    dens[rng.random(dens.shape) < 0.1] = 0.0
    mus = rng.uniform(1, 30, (A, S))
    N = int(rng.integers(1, 300)) if rng.random() < 0.7 else int(rng.integers(1000, 5000))     # (several blocks of 1024 events)
    cols = [rng.uniform(e[0], e[-1], N) for e in edges]
    for ax, e in enumerate(edges):                                             # limits, an inner edge, a bin centre
        special = np.array([e[0], e[-1], e[1], 0.5 * (e[0] + e[1]), 0.5 * (e[-2] + e[-1])])[:N]
        cols[ax][:len(special)] = special
    centres = [0.5 * (e[:-1] + e[1:]) for e in edges]
    grid = edges if method == 'piecewise' else centres
    pts = cols if method == 'piecewise' else [np.clip(c, g[0], g[-1]) for c, g in zip(cols, grid)]
    want = np.empty((A, S, N))
    for a in range(A):
        for s in range(S):
            if method == 'linear':
                want[a, s] = RegularGridInterpolator(centres, dens[a, s])(np.transpose(pts))
            else:
                h = Histdd(bins=edges)
                h.histogram = dens[a, s]
                want[a, s] = h.lookup(*pts)
    tp, ctx = DeviceContext(0), DeviceContext(0)
    try:
        tp.begin_model(anchor_z, S, B)
        for a in range(A):
            tp.set_anchor(a, dens[a].reshape(S, B), mus[a])
        tp.end_model()
        outlier = 1e-12
        tp.score_events(ctx, method, grid, pts, outlier)
        z = np.array([rng.uniform(g[0], g[-1]) for g in anchor_z])
        for a, multi in enumerate(np.ndindex(*[len(g) for g in anchor_z])):     # the scored tensor, anchor by anchor
            za = np.array([g[i] for g, i in zip(anchor_z, multi)])
            got = ctx.interpolate('ps', za).reshape(S, N)
            if method == 'linear' and k == 2:
                np.testing.assert_allclose(got, want[a], rtol=1e-14, atol=1e-300)
            else:
                np.testing.assert_array_equal(got, want[a])
            np.testing.assert_array_equal(ctx.interpolate('mus', za), mus[a])
        model = dict(anchor_z=anchor_z, ps=want.reshape(tuple(len(g) for g in anchor_z) + (S, N)),
                     mus=mus.reshape(tuple(len(g) for g in anchor_z) + (S,)))
        r = rng.uniform(0.2, 2.0, S)
        ll, st = ctx.eval(z if d else None, r)
        ref = orc.loglikelihood_unbinned(model, z, r, outlier)
        assert st[0] == 0 and (ll[0] == ref or abs(ll[0] - ref) <= RTOL * max(1, abs(ref))), (seed, ll[0], ref)
    finally:
        tp.close()
        ctx.close()
