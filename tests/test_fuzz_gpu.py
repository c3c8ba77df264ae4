"""Randomised small configurations against the CPU oracle: every combination of shape-axis count (including
single-anchor axes), source count, ragged bin counts around the 512-bin tile, zero templates, empty bins,
on-anchor / corner points, batch grouping (1..16 points per cell pass), both data forms, gradient, toys."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10


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


@pytest.mark.parametrize('seed', range(12))
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
            one = np.array([ctx.eval(z[i] if d else None, r[i])[0][0] for i in range(min(P, 5))])
            for i in range(P):
                if np.isnan(want[i]) and bb >= 0:               # the reference asserts there
                    assert st[i] & 12, (seed, d, S, B, bb, i)
                    continue
                ok = (got[i] == want[i]) if not np.isfinite(want[i]) else abs(got[i] - want[i]) <= RTOL * max(1, abs(want[i]))
                assert ok, (seed, d, S, B, bb, sparse, maxg, i, got[i], want[i])
            for i in range(len(one)):
                if np.isfinite(want[i]):
                    assert abs(one[i] - want[i]) <= RTOL * max(1, abs(want[i]))
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
