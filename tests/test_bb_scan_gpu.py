"""Beeston-Barlow batches on the fp64 matrix cores (k_scan_bb, csrc/bi_k_scan_bb.h; blueice/likelihood.py:618-660, roots :693-712):
device-planned batches of >= 64 points in which no bin can have U_b == 0 -- against the vector kernel (k_morph_reduce<G, true>,
the same per-bin operations on FMA-interpolated U but reference-order P and a: 1e-12), against the oracle (1e-10), status bits
equal; every variant family (compile-time segments for 8 and 16 corners, run-time segments otherwise); ragged bin counts,
rejected points, several grid cells and datasets; batches in which U_b == 0 is possible never reach the kernel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10
_FIRST, _COUNT = (int(v) for v in os.environ.get('BLUEICE_FUZZ_SEEDS', '0:0').split(':'))


def _model(rng, n_anchor, S, B, T=1):
    shape = tuple(n_anchor)
    anchor_z = [np.sort(rng.uniform(-2, 2, n)) for n in n_anchor]
    ps = rng.random(shape + (S, B)) + 1e-3                          # strictly positive: no bin can have U_b == 0
    ps /= ps.sum(axis=-1, keepdims=True)
    mus = rng.uniform(200, 800, shape + (S,))
    nm = np.ones(shape + (S, B))
    nm[..., 0, :] = 1.0 + rng.poisson(25., shape + (B,))
    lam = (mus.reshape(-1, S)[0][:, None] * ps.reshape(-1, S, B)[0]).sum(axis=0)
    counts = rng.poisson(lam, size=(T, B)).astype(float)
    return dict(anchor_z=anchor_z, ps=ps, mus=mus, n_model=nm), counts


# (S, anchors per axis, bins): 16 corners -> static <4(S-1), 4>; 8 corners -> static <2(S-1), 2>; 4, 2, 1 corners -> run-time segments
SHAPES = [(3, (2, 2, 2, 2), 1000), (6, (2, 3, 2, 2), 517), (2, (3, 2, 2), 4099), (5, (2, 2, 3), 33), (4, (3, 3), 700),
          (3, (4,), 2048), (2, (), 300), (8, (2, 2), 90), (7, (2, 2, 2, 2), 64)]


@pytest.mark.parametrize('S,n_anchor,B', SHAPES)
def test_matrix_core_kernel_equals_the_vector_kernel_and_the_oracle(S, n_anchor, B):
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    rng = np.random.default_rng(17 * S + B)
    T = 2
    model, counts = _model(rng, n_anchor, S, B, T)
    d = len(n_anchor)
    ctx = DeviceContext(0)
    try:
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'], n_model=model['n_model'], bb_source=0)
        ctx.set_param('sparse', 0)
        ctx.upload_counts(counts)
        P = 333
        z = np.stack([rng.uniform(g[0], g[-1], P) for g in model['anchor_z']], axis=1) if d else np.zeros((P, 0))
        r = rng.uniform(0.3, 2.0, (P, S))
        ds = rng.integers(0, T, P)
        if d:
            z[5, 0] = 99.0                                     # outside the box
            z[6] = [g[-1] for g in model['anchor_z']]          # the top corner of the grid
        r[9, 1 % S] = -1.0                                     # unphysical
        ctx.set_param('device_plan_min', 1)                    # the device planner takes the batch whatever its size
        before = ctx.get_param('n_bb_scan_launches')
        got, st = ctx.eval(z if d else None, r, dataset=ds)
        assert ctx.get_param('n_bb_scan_launches') == before + 1
        ctx.set_param('scan_bb', 0)
        vec, st_v = ctx.eval(z if d else None, r, dataset=ds)
        assert ctx.get_param('n_bb_scan_launches') == before + 1
        np.testing.assert_array_equal(st, st_v)
        ok = np.isfinite(vec)
        assert ok.sum() >= P - 2 and np.isneginf(got[~ok]).all()
        np.testing.assert_allclose(got[ok], vec[ok], rtol=1e-12, atol=0)
        for i in (0, 6, 100, P - 1):
            if not ok[i]:
                continue
            want = orc.loglikelihood(model, counts[ds[i]], z[i], r[i], bb_source=0)
            assert st[i] == 0 and abs(got[i] - want) <= RTOL * max(1.0, abs(want)), (i, got[i], want)
        # a batch below scan_bb_min keeps the vector kernel; the same points come out the same (to rounding)
        ctx.set_param('scan_bb', 1)
        few, _ = ctx.eval(z[:40] if d else None, r[:40], dataset=ds[:40])
        assert ctx.get_param('n_bb_scan_launches') == before + 1
        np.testing.assert_allclose(few[ok[:40]], got[:40][ok[:40]], rtol=1e-12, atol=0)
    finally:
        ctx.close()


def test_batches_with_possible_zero_u_never_reach_the_matrix_core_kernel():
    """Where some bin can have U_b == 0 the reference's first-root assertion hangs on the last bits of P, a and N
    (tests/test_bb_exact_gpu.py): such batches are the host planner's (exact totals, reference-order interpolation), whatever
    their size -- the values and assertion bits of large batches equal those of single calls."""
    from blueice_amd.device import DeviceContext
    rng = np.random.default_rng(5)
    model, counts = _model(rng, (2, 2, 2), 3, 900)
    model['ps'][..., 1:, 17] = 0.0                              # one bin in which only the Beeston-Barlow source expects anything
    model['ps'] /= model['ps'].sum(axis=-1, keepdims=True)
    ctx = DeviceContext(0)
    try:
        ctx.upload_model(model['anchor_z'], model['ps'], model['mus'], n_model=model['n_model'], bb_source=0)
        ctx.set_param('sparse', 0)
        ctx.upload_counts(counts)
        P = 200
        z = np.stack([rng.uniform(g[0], g[-1], P) for g in model['anchor_z']], axis=1)
        r = rng.uniform(0.3, 2.0, (P, 3))
        ctx.set_param('device_plan_min', 1)
        before = ctx.get_param('n_bb_scan_launches')
        got, st = ctx.eval(z, r)
        assert ctx.get_param('n_bb_scan_launches') == before
        for i in range(0, P, 17):
            one, st1 = ctx.eval(z[i], r[i])
            assert st1[0] == st[i]
            if np.isfinite(one[0]):
                assert one[0] == got[i]
    finally:
        ctx.close()


@pytest.mark.parametrize('seed', range(_FIRST, _FIRST + _COUNT) if _COUNT else range(6))
def test_random_models_against_the_oracle(seed):
    """Random shapes (0 ... 4 shape axes of 1 ... 3 anchors, 2 ... 8 sources, any Beeston-Barlow source, ragged bin counts, 1 ... 3
    datasets, small and large Monte-Carlo counts, switched-off sources, points on anchors and on the grid's top edge) through the
    device planner and k_scan_bb: EVERY point against the oracle, status bits included (BLUEICE_FUZZ_SEEDS=first:count)."""
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    rng = np.random.default_rng(23000 + seed)
    d = int(rng.integers(0, 5))
    n_anchor = [int(rng.integers(1, 4)) for _ in range(d)]
    S = int(rng.integers(2, 9))
    B = int(rng.choice([1, 15, 16, 17, 90, 512, 700, 1300]))
    T = int(rng.integers(1, 4))
    bb = int(rng.integers(0, S))
    shape = tuple(n_anchor)
    anchor_z = [np.sort(rng.uniform(-2, 2, n)) for n in n_anchor]
    ps = rng.random(shape + (S, B)) ** 2 + 1e-3                     # strictly positive: no bin can have U_b == 0
    ps /= ps.sum(axis=-1, keepdims=True)
    mus = rng.uniform(5, 60, shape + (S,)) * rng.choice([1.0, 10.0])
    nm = np.ones(shape + (S, B))
    nm[..., bb, :] = rng.choice([0.0, 1.0]) + rng.poisson(rng.choice([0.7, 25.0]), shape + (B,))    # (also bins without Monte-Carlo events)
    model = dict(anchor_z=anchor_z, ps=ps, mus=mus, n_model=nm)
    lam = (mus.reshape(-1, S)[0][:, None] * ps.reshape(-1, S, B)[0]).sum(axis=0)
    counts = rng.poisson(lam, size=(T, B)).astype(float)
    P = int(rng.integers(64, 200))
    z = np.zeros((P, d))
    for i, g in enumerate(anchor_z):
        mode = rng.integers(0, 5, P)
        z[:, i] = np.where(mode == 0, rng.choice(g, P), np.where(mode == 1, g[-1], rng.uniform(g[0], g[-1], P)))
    r = rng.uniform(0.3, 2.0, (P, S))
    off = rng.random((P, S)) < 0.08
    off[:, bb] = False                                               # (the Beeston-Barlow source switched off: p_cal = 0, left to the vector kernels' tests)
    off[np.arange(P), (bb + 1) % S] = False                          # one other source stays on: U_b > 0 in every bin
    r[off] = 0.0
    ds = rng.integers(0, T, P)
    if d:
        z[5, 0] = 99.0
    r[9, (bb + 1) % S] = -1.0
    ctx = DeviceContext(0)
    try:
        ctx.upload_model(anchor_z, ps, mus, n_model=nm, bb_source=bb)
        ctx.set_param('sparse', 0)
        ctx.upload_counts(counts)
        ctx.set_param('device_plan_min', 1)
        before = ctx.get_param('n_bb_scan_launches')
        got, st = ctx.eval(z if d else None, r, dataset=ds)
        ran = ctx.get_param('n_bb_scan_launches') - before
        for i in range(P):
            with np.errstate(all='ignore'):
                try:
                    want = orc.loglikelihood(model, counts[ds[i]], z[i], r[i], bb_source=bb)
                except AssertionError:
                    assert st[i] & 12, (seed, i, st[i], got[i])      # the reference's assertions on the roots: reported as status bits
                    continue
            g = got[i]
            ok = (np.isnan(want) and np.isnan(g)) or g == want or (np.isfinite(want) and abs(g - want) <= RTOL * max(1.0, abs(want)))
            assert ok, (seed, d, n_anchor, S, B, T, bb, ran, i, g, want, st[i])
        print('k_scan_bb launches: %d' % ran)
    finally:
        ctx.close()
