"""Analytic gradient (bi_eval_grad / value_and_gradient): against central finite differences of the CPU
oracle on the same inputs, in both the dense and the non-empty-bin form, and through bestfit_scipy."""
import numpy as np
import pytest

import model_zoo
from golden_util import GOLDEN_DIR, load_case

pytestmark = pytest.mark.gpu


def fd_oracle(model, counts, z, r, h=1e-6):
    from oracle import blueice_oracle as orc
    gz = np.zeros(len(z))
    for i in range(len(z)):
        zp, zm = np.array(z, float), np.array(z, float)
        zp[i] += h
        zm[i] -= h
        gz[i] = (orc.loglikelihood(model, counts, zp, r) - orc.loglikelihood(model, counts, zm, r)) / (2 * h)
    gr = np.zeros(len(r))
    for s in range(len(r)):
        hs = h * max(1.0, abs(r[s]))
        rp, rm = np.array(r, float), np.array(r, float)
        rp[s] += hs
        rm[s] -= hs
        gr[s] = (orc.loglikelihood(model, counts, z, rp) - orc.loglikelihood(model, counts, z, rm)) / (2 * hs)
    return gz, gr


@pytest.mark.parametrize('sparse', [0, 2])
@pytest.mark.parametrize('name', ['c1_like', 'd2_nonuniform', 'd3_small', 'd0_multi_source'])
def test_gradient_matches_finite_differences(name, sparse):
    from blueice_amd.device import DeviceContext
    from oracle import blueice_oracle as orc
    c = load_case(name)
    ctx = DeviceContext(0)
    ctx.set_param('sparse', sparse)
    ctx.upload_model(c['model']['anchor_z'], c['model']['ps'], c['model']['mus'])
    ctx.upload_counts(c['counts'])
    rng = np.random.default_rng(3)
    zs = np.array([[rng.uniform(g[0], g[-1]) for g in c['model']['anchor_z']] for _ in range(6)]).reshape(6, c['d'])
    rs = rng.uniform(0.4, 1.6, size=(6, c['S']))
    ll, gz, gs, st = ctx.eval_grad(zs if c['d'] else None, rs)
    ref, _ = ctx.eval(zs if c['d'] else None, rs)
    assert not st.any()
    np.testing.assert_allclose(ll, ref, rtol=1e-12)
    for i in range(6):
        fz, fr = fd_oracle(c['model'], c['counts'], zs[i], rs[i])
        scale = max(1.0, np.abs(np.concatenate([fz, fr])).max())
        np.testing.assert_allclose(gz[i], fz, atol=2e-5 * scale, rtol=1e-6)
        np.testing.assert_allclose(gs[i], fr, atol=2e-5 * scale, rtol=1e-6)
    # status / edge: out of the box -> -inf with NaN gradient
    if c['d']:
        zz = zs[0].copy()
        zz[0] = c['model']['anchor_z'][0][-1] + 1.0
        ll, gz, gs, st = ctx.eval_grad(zz, rs[0])
        assert ll[0] == -np.inf and st[0] == 1 and np.isnan(gz[0]).all()
    ctx.close()


def test_value_and_gradient_through_the_likelihood_class():
    from scipy import stats
    ns = model_zoo.namespace_of('blueice_amd')
    lf, _, _ = model_zoo.d2_nonuniform(ns)
    lf.add_rate_uncertainty('s1', 0.3)
    kw = dict(shift=0.3, stretch=2.2, s0_rate_multiplier=0.8, s1_rate_multiplier=1.2, s2_rate_multiplier=1.1,
              livetime_days=3.)
    ll, grads = lf.value_and_gradient(**kw)
    assert abs(ll - lf(**kw)) <= 1e-12 * abs(ll)
    assert list(grads) == ['s0_rate_multiplier', 's1_rate_multiplier', 's2_rate_multiplier', 'shift', 'stretch']
    for name, g in grads.items():
        h = 1e-6
        up, dn = dict(kw), dict(kw)
        up[name] += h
        dn[name] -= h
        fd = (lf(**up) - lf(**dn)) / (2 * h)
        assert abs(g - fd) <= 1e-5 * max(1.0, abs(fd)), (name, g, fd)


def test_bestfit_with_gradient_reaches_the_reference_optimum():
    f = np.load(GOLDEN_DIR + '/fit_c1_like.npz')
    ns = model_zoo.namespace_of('blueice_amd')
    lf = model_zoo.fit_c1_like(ns)
    calls = {'n': 0}
    orig = lf.ctx.eval_grad

    def counting(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    lf.ctx.eval_grad = counting
    res, ll = lf.bestfit_scipy(use_gradient=True)
    assert list(res.keys()) == [str(x) for x in f['fit_all_names']]
    assert ll >= float(f['fit_all_ll']) - 1e-6 * abs(ll)           # at least as good as the reference's optimum
    assert abs(ll - float(f['fit_all_ll'])) < 1e-4 * abs(ll)
    assert 0 < calls['n'] < 200                                      # vs ~500 objective calls by differencing


def test_gradient_with_efficiency_parameter():
    """A shape parameter that also scales the rates of some sources (efficiency) enters the gradient twice."""
    ns = model_zoo.namespace_of('blueice_amd')
    lf, _, _ = model_zoo.efficiency_param(ns)
    kw = dict(eff=1.2, shift=0.3, a_rate_multiplier=0.9, b_rate_multiplier=1.4, c_rate_multiplier=0.6)
    ll, grads = lf.value_and_gradient(**kw)
    assert abs(ll - lf(**kw)) <= 1e-12 * abs(ll)
    for name, g in grads.items():
        h = 1e-6
        up, dn = dict(kw), dict(kw)
        up[name] += h
        dn[name] -= h
        fd = (lf(**up) - lf(**dn)) / (2 * h)
        assert abs(g - fd) <= 2e-5 * max(1.0, abs(fd)), (name, g, fd)


# ---- Beeston-Barlow: the chain rule through the per-bin root (blueice/likelihood.py:618-660,693-712)
def fd_oracle_bb(model, counts, z, r, bb, h=1e-6):
    from oracle import blueice_oracle as orc
    f = lambda zz, rr: orc.loglikelihood(model, counts, zz, rr, bb_source=bb, forgive_zero_u=True)
    gz = np.zeros(len(z))
    for i in range(len(z)):
        zp, zm = np.array(z, float), np.array(z, float)
        zp[i] += h
        zm[i] -= h
        gz[i] = (f(zp, r) - f(zm, r)) / (2 * h)
    gr = np.zeros(len(r))
    for s in range(len(r)):
        hs = h * max(1.0, abs(r[s]))
        rp, rm = np.array(r, float), np.array(r, float)
        rp[s] += hs
        rm[s] -= hs
        gr[s] = (f(z, rp) - f(z, rm)) / (2 * hs)
    return gz, gr


@pytest.mark.parametrize('name', ['ref_bb_second_source', 'bb_two_shape', 'bb_d2'])
def test_beeston_barlow_gradient_matches_finite_differences(name):
    from blueice_amd.device import DeviceContext
    c = load_case(name)
    bb = c['bb_source']
    ctx = DeviceContext(0)
    ctx.upload_model(c['model']['anchor_z'], c['model']['ps'], c['model']['mus'], n_model=c['model']['n_model'], bb_source=bb)
    ctx.upload_counts(c['counts'])
    rng = np.random.default_rng(5)
    n = 6
    zs = np.array([[rng.uniform(g[0] + 0.02 * (g[-1] - g[0]), g[-1] - 0.02 * (g[-1] - g[0])) for g in c['model']['anchor_z']]
                   for _ in range(n)]).reshape(n, c['d'])
    rs = rng.uniform(0.5, 1.6, size=(n, c['S']))
    ll, gz, gs, st = ctx.eval_grad(zs if c['d'] else None, rs)
    ref, rst = ctx.eval(zs if c['d'] else None, rs)
    np.testing.assert_array_equal(st, rst)                    # the assertion bits of the value ride along
    ok = st == 0
    assert ok.sum() >= 3
    np.testing.assert_allclose(ll[ok], ref[ok], rtol=1e-13)
    for i in np.flatnonzero(ok):
        fz, fr = fd_oracle_bb(c['model'], c['counts'], zs[i], rs[i], bb)
        scale = max(1.0, np.abs(np.concatenate([fz, fr])).max())
        np.testing.assert_allclose(gz[i], fz, atol=1e-6 * scale, rtol=2e-6)
        np.testing.assert_allclose(gs[i], fr, atol=1e-6 * scale, rtol=2e-6)
    ctx.close()


def test_beeston_barlow_gradient_synthetic_mini_and_zero_u_bins():
    """mini4bb (4 shape axes, one of them a single anchor; 3 sources -> 8 columns, DZ = 8) and a model in which the other
    sources expect exactly nothing in some bins (U_b == 0: the reference's special case, differentiated as such)."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini4bb', bb_source=0)
    dense = m.dense_model()
    counts = m.counts(dense=True)
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.upload_counts(counts)
    z, r = m.random_points(5, seed=3)
    ll, gz, gs, st = ctx.eval_grad(z, r)
    assert not st.any()
    np.testing.assert_allclose(ll, ctx.eval(z, r)[0], rtol=1e-13)
    for i in range(5):
        fz, fr = fd_oracle_bb(dense, counts, z[i], r[i], 0)
        scale = max(1.0, np.abs(np.concatenate([fz, fr])).max())
        np.testing.assert_allclose(gz[i], fz, atol=1e-6 * scale, rtol=2e-6)
        np.testing.assert_allclose(gs[i], fr, atol=1e-6 * scale, rtol=2e-6)
    # exact zeros of the other sources' templates in a block of bins
    shape = dense['ps'].shape
    ps = dense['ps'].reshape(shape[:m.d + 1] + (m.B,)).copy()
    ps[..., 1:, 100:160] = 0.0
    ps = ps.reshape(shape)
    model = dict(dense, ps=ps)
    ctx.upload_model(model['anchor_z'], ps, model['mus'], n_model=model['n_model'], bb_source=0)
    ctx.upload_counts(counts)
    ll, gz, gs, st = ctx.eval_grad(z, r)
    ref, rst = ctx.eval(z, r)
    np.testing.assert_array_equal(st & ~4, rst & ~4)
    for i in range(5):
        fz, fr = fd_oracle_bb(model, counts, z[i], r[i], 0)
        scale = max(1.0, np.abs(np.concatenate([fz, fr])).max())
        np.testing.assert_allclose(gz[i], fz, atol=2e-6 * scale, rtol=5e-6)
        np.testing.assert_allclose(gs[i], fr, atol=2e-6 * scale, rtol=5e-6)
    ctx.close()


def test_beeston_barlow_fit_with_analytic_gradient():
    """bestfit_scipy(use_gradient=True) on a Beeston-Barlow likelihood reaches the optimum of the differencing fit in a
    fraction of the device calls; the batched engine profiles it too."""
    ns = model_zoo.namespace_of('blueice_amd')
    lf, _, _ = model_zoo.bb_d2(ns)
    assert lf.supports_gradient
    kw = {k: v for k, v in zip(lf.shape_parameters, [0.3, 0.4])}
    ll, grads = lf.value_and_gradient(**kw)
    assert abs(ll - lf(**kw)) <= 1e-12 * abs(ll)
    for name, g in grads.items():
        h = 1e-6
        up, dn = dict(kw), dict(kw)
        up[name] = up.get(name, 1.0) + h
        dn[name] = dn.get(name, 1.0) - h
        fd = (lf(**up) - lf(**dn)) / (2 * h)
        assert abs(g - fd) <= 2e-5 * max(1.0, abs(fd)), (name, g, fd)
    lf.ctx.set_param('single_timing_reset', 1)
    res_fd, ll_fd = lf.bestfit_scipy(batch_stencil=False)
    scalar_calls = lf.ctx.get_param('single_calls')
    calls = {'n': 0}
    orig = lf.ctx.eval_grad

    def counting(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    lf.ctx.eval_grad = counting
    res_g, ll_g = lf.bestfit_scipy(use_gradient=True)
    assert ll_g >= ll_fd - 1e-6 * abs(ll_fd)
    assert abs(ll_g - ll_fd) <= 1e-4 * abs(ll_fd)
    assert 0 < calls['n'] < scalar_calls / 2
    best, ll_b, info = lf.bestfit_batched(return_info=True)
    assert info['analytic_gradient'] and ll_b[0] >= ll_fd - 1e-6 * abs(ll_fd)


@pytest.mark.parametrize('sparse', [0, 2])
@pytest.mark.parametrize('name', ['c1_like', 'd2_nonuniform', 'd3_small', 'd0_multi_source'])
def test_large_gradient_batches_planned_on_the_device(name, sparse):
    """bi_eval_grad builds the descriptors of batches of >= 512 points on the device (k_grad_fill): same values, slopes
    and status words as the host-planned path, rejected points (outside the box, nan, unphysical rates, bad dataset)
    and several datasets included."""
    from blueice_amd.device import DeviceContext
    c = load_case(name)
    ctx = DeviceContext(0)
    ctx.set_param('sparse', sparse)
    ctx.upload_model(c['model']['anchor_z'], c['model']['ps'], c['model']['mus'])
    rng = np.random.default_rng(9)
    counts = np.stack([c['counts'], rng.poisson(c['counts'] + 0.3).astype(float), np.zeros_like(c['counts'])])
    ctx.upload_counts(counts)
    P = 900
    zs = np.array([[rng.uniform(g[0], g[-1]) for g in c['model']['anchor_z']] for _ in range(P)]).reshape(P, c['d'])
    rs = rng.uniform(0.4, 1.6, size=(P, c['S']))
    ds = rng.integers(0, 3, P)
    if c['d']:
        zs[3, 0] = c['model']['anchor_z'][0][-1] + 1.0
        zs[4, c['d'] - 1] = np.nan
        zs[5] = [g[-1] for g in c['model']['anchor_z']]          # the top corner of the box
        zs[6] = [g[0] for g in c['model']['anchor_z']]
    rs[7, 0] = -0.5
    rs[8] = 0.0
    ds[9] = 11
    got = ctx.eval_grad(zs if c['d'] else None, rs, ds)
    ctx.set_param('device_plan_min', 0)                          # the same call planned on the host
    want = ctx.eval_grad(zs if c['d'] else None, rs, ds)
    np.testing.assert_array_equal(got[3], want[3])
    assert got[3][7] == 2 and got[3][9] == 16 and (not c['d'] or (got[3][3] == 1 and got[3][4] == 1))
    ok = want[3] == 0
    assert ok.sum() > P - 10
    np.testing.assert_allclose(got[0][ok], want[0][ok], rtol=1e-13)
    assert np.all(np.isneginf(got[0][~ok])) and np.all(np.isnan(got[1][~ok])) and np.all(np.isnan(got[2][~ok]))
    scale = np.maximum(1.0, np.abs(want[0][ok]))[:, None]
    np.testing.assert_allclose(got[1][ok], want[1][ok], rtol=1e-10, atol=1e-11 * scale.max())
    np.testing.assert_allclose(got[2][ok], want[2][ok], rtol=1e-10, atol=1e-11 * scale.max())
    ctx.close()


# ---- the extended unbinned likelihood (bi_eval_grad in MODE 3; VERDICT round 3, "Next round" 5) -------------------------
def _unbinned_context(c):
    from blueice_amd.device import DeviceContext
    ctx = DeviceContext(0)
    n_ev = c['bins'][0]
    grid_shape = tuple(len(g) for g in c['model']['anchor_z'])
    ctx.begin_model(c['model']['anchor_z'], c['S'], n_ev)
    ps = c['model']['ps'].reshape((-1, c['S'], n_ev)) if n_ev else np.zeros((int(np.prod(grid_shape)), c['S'], 0))
    mus = c['model']['mus'].reshape((-1, c['S']))
    for a in range(len(mus)):
        ctx.set_anchor(a, ps[a], mus[a])
    ctx.end_model()
    ctx.set_unbinned(c['outlier'])
    return ctx


def fd_oracle_unbinned(model, z, r, outlier, h=1e-6):
    from oracle import blueice_oracle as orc
    f = lambda zz, rr: orc.loglikelihood_unbinned(model, zz, rr, outlier)
    gz = np.zeros(len(z))
    for i in range(len(z)):
        zp, zm = np.array(z, float), np.array(z, float)
        zp[i] += h
        zm[i] -= h
        gz[i] = (f(zp, r) - f(zm, r)) / (2 * h)
    gr = np.zeros(len(r))
    for s in range(len(r)):
        hs = h * max(1.0, abs(r[s]))
        rp, rm = np.array(r, float), np.array(r, float)
        rp[s] += hs
        rm[s] -= hs
        gr[s] = (f(z, rp) - f(z, rm)) / (2 * hs)
    return gz, gr


@pytest.mark.parametrize('name', ['unb_ref_value', 'unb_shape_2src', 'unb_d0_three_sources', 'unb_no_events', 'unb_nan_pdf', 'unb_mc_hist'])
def test_unbinned_gradient_matches_finite_differences_of_the_oracle(name):
    """d ll = -sum_s d mu_s + sum_e (sum_s d(mu_s p_s)) / (sum_s mu_s p_s): against central differences of
    orc.loglikelihood_unbinned at 1e-6 on the reference's unbinned fixtures -- events on the outlier clamp contribute no
    slope, nan pdf terms are dropped from value and slopes alike (np.nansum, blueice/likelihood.py:686)."""
    from golden_util import load_case
    c = load_case(name)
    ctx = _unbinned_context(c)
    rng = np.random.default_rng(8)
    n = 6
    zs = np.array([[rng.uniform(g[0] + 0.02 * (g[-1] - g[0]), g[-1] - 0.02 * (g[-1] - g[0])) if len(g) > 1 else g[0]
                    for g in c['model']['anchor_z']] for _ in range(n)]).reshape(n, c['d'])
    rs = rng.uniform(0.4, 1.6, size=(n, c['S']))
    ll, gz, gs, st = ctx.eval_grad(zs if c['d'] else None, rs)
    ref, _ = ctx.eval(zs if c['d'] else None, rs)
    assert not st.any()
    np.testing.assert_allclose(ll, ref, rtol=1e-13)                   # the value column IS bi_eval's value
    for i in range(n):
        fz, fr = fd_oracle_unbinned(c['model'], zs[i], rs[i], c['outlier'])
        scale = max(1.0, np.abs(np.concatenate([fz, fr])).max())
        np.testing.assert_allclose(gz[i], fz, atol=1e-6 * scale, rtol=1e-6, err_msg='%s point %d: shape slopes' % (name, i))
        np.testing.assert_allclose(gs[i], fr, atol=1e-6 * scale, rtol=1e-6, err_msg='%s point %d: rate slopes' % (name, i))
    if c['d']:                                                        # outside the anchor box: -inf, nan slopes
        zz = zs[0].copy()
        zz[0] = c['model']['anchor_z'][0][-1] + 1.0
        ll, gz, gs, st = ctx.eval_grad(zz, rs[0])
        assert ll[0] == -np.inf and st[0] == 1 and np.isnan(gz[0]).all() and np.isnan(gs[0]).all()
    ctx.close()


def test_unbinned_likelihood_class_has_an_analytic_gradient():
    """UnbinnedLogLikelihood.supports_gradient: value_and_gradient / values_and_gradients through the class (priors, live
    time), bestfit_scipy(use_gradient=True) reaching the plain fit's maximum."""
    ns = model_zoo.namespace_of('blueice_amd')
    lf, calls, _ = model_zoo.UNBINNED_CASES['unb_shape_2src'](ns)
    assert lf.supports_gradient
    names = list(lf.shape_parameters)
    lo, hi = lf.get_bounds(names[0])
    kw = {names[0]: lo + 0.37 * (hi - lo), 's0_rate_multiplier': 1.3}
    ll, grads = lf.value_and_gradient(**kw)
    assert abs(ll - lf(**kw)) <= 1e-12 * max(1.0, abs(ll))
    for k in grads:
        h = 1e-6 * max(1.0, abs(kw.get(k, 1.0)))
        up, dn = dict(kw), dict(kw)
        up[k] = kw.get(k, 1.0) + h
        dn[k] = kw.get(k, 1.0) - h
        fd = (lf(**up) - lf(**dn)) / (2 * h)
        assert abs(grads[k] - fd) <= 1e-5 * max(1.0, abs(fd)), (k, grads[k], fd)
    pts = {names[0]: np.linspace(lo, hi, 9)[1:-1], 's0_rate_multiplier': 1.3}
    v, g = lf.values_and_gradients(pts)
    np.testing.assert_allclose(v, lf.eval_points(pts), rtol=1e-13)
    best, top = lf.bestfit_scipy()
    best_g, top_g = lf.bestfit_scipy(use_gradient=True)
    assert abs(top_g - top) <= 1e-6 * max(1.0, abs(top)) or top_g > top


# ---- large batches on the matrix cores (k_grad_mfma; VERDICT round 3, "Next round" 4) ----------------------------------
@pytest.mark.parametrize('sparse', [0, 2])
@pytest.mark.parametrize('name', ['mini3', 'c1_like', 'd2_nonuniform', 'd3_small', 'd0_multi_source'])
def test_matrix_core_gradient_equals_one_work_item_per_point(name, sparse):
    """bi_eval_grad over >= 512 points of one dataset: mu = rows x coefficients and G = (n / mu) x rows^T as two chained
    matrix products per 16-bin block, the derivative coefficients contracted per point afterwards -- against the path with
    one work item per point (grad_mfma = 0), against central differences of the oracle, with rejected points in the batch."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    ctx = DeviceContext(0)
    if name == 'mini3':
        m = SyntheticModel.named('mini3')
        m.upload(ctx)
        anchor_z, S, counts, model = m.anchor_z, m.S, m.counts(dense=bool(sparse == 0)), m.dense_model()
        ctx.set_param('sparse', sparse)
        ctx.upload_counts(counts)
    else:
        c = load_case(name)
        ctx.set_param('sparse', sparse)
        ctx.upload_model(c['model']['anchor_z'], c['model']['ps'], c['model']['mus'])
        ctx.upload_counts(c['counts'])
        anchor_z, S, counts, model = c['model']['anchor_z'], c['S'], c['counts'], c['model']
    d = len(anchor_z)
    rng = np.random.default_rng(11)
    P = 1700
    zs = np.array([[rng.uniform(g[0], g[-1]) if len(g) > 1 else g[0] for g in anchor_z] for _ in range(P)]).reshape(P, d)
    rs = rng.uniform(0.3, 1.7, size=(P, S))
    if d:
        zs[5, 0] = anchor_z[0][-1] + 1.0          # outside the box
        zs[6] = [g[-1] for g in anchor_z]          # the top corner of the grid
    rs[7, 0] = -0.5                                # an unphysical rate
    ctx.set_param('grad_mfma_min', 512)            # (by default the path starts at 2048 points, where it pays)
    before = ctx.get_param('n_grad_mfma_launches')
    ll, gz, gs, st = ctx.eval_grad(zs if d else None, rs)
    assert ctx.get_param('n_grad_mfma_launches') == before + 1, 'the batch did not take the matrix-core gradient kernel'
    ctx.set_param('grad_mfma', 0)
    ll0, gz0, gs0, st0 = ctx.eval_grad(zs if d else None, rs)
    assert ctx.get_param('n_grad_mfma_launches') == before + 1
    ctx.set_param('grad_mfma', 1)
    np.testing.assert_array_equal(st, st0)
    assert st[7] == 2 and ll[7] == -np.inf and np.isnan(gs[7]).all()
    if d:
        assert st[5] == 1 and ll[5] == -np.inf and np.isnan(gz[5]).all()
    ok = st == 0
    assert ok.sum() >= P - 3
    np.testing.assert_allclose(ll[ok], ll0[ok], rtol=1e-12)
    scale = np.maximum(1.0, np.abs(np.concatenate([gz0[ok], gs0[ok]], axis=1)).max(axis=1))[:, None]
    np.testing.assert_allclose(gz[ok] / scale, gz0[ok] / scale, atol=1e-9)
    np.testing.assert_allclose(gs[ok] / scale, gs0[ok] / scale, atol=1e-9)
    ref, _ = ctx.eval(zs if d else None, rs)
    np.testing.assert_allclose(ll[ok], ref[ok], rtol=1e-12)
    for i in (0, 6, 333, 1699):
        if not ok[i]:
            continue
        fz, fr = fd_oracle(model, counts, zs[i], rs[i])
        sc = max(1.0, np.abs(np.concatenate([fz, fr])).max())
        if i != 6:                                  # (on the grid's corner the central difference leaves the box)
            np.testing.assert_allclose(gz[i], fz, atol=2e-5 * sc, rtol=1e-6)
        np.testing.assert_allclose(gs[i], fr, atol=2e-5 * sc, rtol=1e-6)
    ctx.close()


def test_matrix_core_gradient_keeps_scipys_edge_cases():
    """Counts that are nan / negative / non-integer, expectations of zero where there is data: the value column must be what
    bi_eval gives (nan / -inf), and the slopes of such points nan."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.set_param('sparse', 0)
    z, r = m.random_points(900, seed=4)
    ctx.set_param('grad_mfma_min', 512)
    for kind in ('nan', 'negative', 'half'):
        counts = m.counts(dense=True).copy()
        counts[37] = {'nan': np.nan, 'negative': -2.0, 'half': 0.5}[kind]
        ctx.upload_counts(counts)
        ll, gz, gs, st = ctx.eval_grad(z, r)
        ref, _ = ctx.eval(z, r)
        if kind == 'nan':
            assert np.isnan(ll).all() and np.isnan(ref).all()
        else:
            assert (ll == -np.inf).all() and (ref == -np.inf).all()
        assert np.isnan(gz).all() and np.isnan(gs).all()
    ctx.close()
