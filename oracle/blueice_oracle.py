"""CPU oracle: a numpy/scipy.special restatement of blueice's binned-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `blueice_amd/` may import this module; the only
legal importers are `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` -- and there only as the checker / the timed CPU baseline, never as the product.

What is restated (file:line are into the reference tree, JelleAalbers/blueice v1.2.1):

  a3  GridInterpolator.make_interpolator's closure `lambda zs: itp(zs)[0]`
      (blueice/pdf_morphers.py:57-70) == scipy.interpolate.RegularGridInterpolator
      (method='linear', bounds_error=True).  The arithmetic lives in scipy (requirement
      `scipy>=0.15`, unpinned; this oracle is pinned against scipy 1.15.3
      `_rgi.py:_find_indices/_evaluate_linear`): per axis the interval
      g[k] <= z < g[k+1] (last interval closed), t=(z-g[k])/(g[k+1]-g[k]); the value is
      sum over corners in itertools.product order (axis 0 slowest) of
      V[corner] * (((1*w_0)*w_1)...), accumulated left to right from 0.0.
  a4  the rate pipeline r_s = mus_interp(z)_s * rate_scale_s (blueice/likelihood.py:355,366-393)
      and its early exits (-inf outside the anchor box :345-347, unphysical rates :397-415).
  a5  BinnedLogLikelihood._compute_likelihood (blueice/likelihood.py:662-675):
      mu_b = ((e_0+e_1)+e_2)+..., e_s = p_s*r_s; term_b = (xlogy(n,mu) - gammaln(n+1)) - mu
      with scipy.stats.poisson.logpmf's argument/support handling; logL = np.sum(term).
  a6  BinnedLogLikelihood.adjust_expectations, 'bb_single' (blueice/likelihood.py:618-660)
      with beeston_barlow_root1/2 (:693-712).
  (next row f-4) extended_loglikelihood (blueice/likelihood.py:678-690) behind
      UnbinnedLogLikelihood._compute_likelihood (:571-573): -sum mu + sum_e log(nansum_s mu_s p_s(x_e))
      with the outlier clamp.

Pinning: tests/test_oracle_golden.py checks every function here against
  * the golden fixtures in tests/golden/*.npz, which were produced by importing the real
    reference in the development container (tests/golden/make_golden.py), and
  * the closed-form expectations of the reference's own tests
    (tests/test_binned_likelihood.py, tests/test_BeestonBarlow.py, test_likelihood.py::test_zero_bin).
"""
import itertools

import numpy as np
from scipy.special import gammaln, xlogy

__all__ = ['find_cell', 'corner_terms', 'interpolate', 'poisson_logpmf', 'compute_likelihood',
           'beeston_barlow_root1', 'beeston_barlow_root2', 'adjust_expectations_bb',
           'in_bounds', 'rates_at', 'loglikelihood', 'loglikelihood_batch',
           'extended_loglikelihood', 'loglikelihood_unbinned']


def find_cell(grid, z):
    """Interval index k and normalised distance t on one axis (scipy `find_indices`).

    g[k] <= z < g[k+1]; z == g[-1] belongs to the last interval (t = 1).  A size-1 axis
    returns (0, 0.0): the "upper" corner then aliases the lower one with weight exactly 0."""
    g = np.asarray(grid, dtype=float)
    n = len(g)
    if n == 1:
        return 0, 0.0
    if z == g[-1]:
        k = n - 2
    else:
        k = int(np.searchsorted(g, z, side='right')) - 1
        k = min(max(k, 0), n - 2)
    denom = g[k + 1] - g[k]
    t = (z - g[k]) / denom
    return k, float(t)


def corner_terms(anchor_z_arrays, z):
    """List of (anchor multi-index, weight) in scipy's hypercube order.

    weight = ((1.0 * w_0) * w_1) ... with w_i = (1 - t_i) for the lower and t_i for the upper
    corner on axis i (scipy `_evaluate_linear`)."""
    cells = [find_cell(g, zi) for g, zi in zip(anchor_z_arrays, z)]
    per_axis = []
    for g, (k, t) in zip(anchor_z_arrays, cells):
        lo = (k, 1 - t)
        hi = (k + 1 if len(g) > 1 else k, t)
        per_axis.append((lo, hi))
    out = []
    for h in itertools.product(*per_axis):
        idx, ws = zip(*h) if h else ((), ())
        weight = 1.0
        for w in ws:
            weight = weight * w
        out.append((tuple(idx), weight))
    return out


def interpolate(anchor_z_arrays, values, z):
    """Multilinear interpolation of the anchor tensor `values[A_0..A_{d-1}, *extra]` at z[d]: what the closure
    returned by GridInterpolator.make_interpolator computes (blueice/pdf_morphers.py:67-70, i.e. scipy
    RegularGridInterpolator.__call__ with method='linear', bounds_error=True; `_rgi.py:_evaluate_linear`)."""
    values = np.asarray(values)
    z = np.asarray(z, dtype=float)
    d = len(anchor_z_arrays)
    for g, zi in zip(anchor_z_arrays, z):
        if not (g[0] <= zi <= g[-1]):
            raise ValueError("One of the requested xi is out of bounds")
    value = np.zeros(values.shape[d:], dtype=float)
    for idx, weight in corner_terms(anchor_z_arrays, z):
        value = value + values[idx] * weight
    return value


def poisson_logpmf(k, mu):
    """scipy.stats.poisson(mu).logpmf(k), restated (scipy 1.15.3 `_discrete_distns.py:996-998`,
    `_distn_infrastructure.py:3503-3539`): nan where mu is not >= 0 (or k is nan);
    -inf where k is negative or non-integer; else (xlogy(k, mu) - gammaln(k + 1)) - mu."""
    k = np.asarray(k, dtype=float)
    mu = np.asarray(mu, dtype=float)
    k, mu = np.broadcast_arrays(k, mu)
    cond0 = mu >= 0
    cond1 = (k >= 0) & (np.floor(k) == k)
    out = np.full(k.shape, -np.inf)
    good = cond0 & cond1
    with np.errstate(all='ignore'):
        out[good] = (xlogy(k[good], mu[good]) - gammaln(k[good] + 1)) - mu[good]
    out[~cond0 | np.isnan(k)] = np.nan
    return out


def compute_likelihood(mus, pmfs, counts):
    """a5: BinnedLogLikelihood._compute_likelihood (blueice/likelihood.py:662-675):
    sum_b poisson.logpmf(n_b | sum_s mus_s pmfs_{s,b}), rows scaled in place then np.sum(axis=0)."""
    expected = np.array(pmfs, dtype=float, copy=True)
    for mu, row in zip(mus, expected):
        row *= mu
    total = np.sum(expected, axis=0)
    return np.sum(poisson_logpmf(counts, total))


def _bb_disc(a, p, U, d):
    return (U**2*p**2 + 2*U**2*p + U**2 + 2*U*a*p**2 + 2*U*a*p -
            2*U*d*p**2 - 2*U*d*p + a**2*p**2 + 2*a*d*p**2 + d**2*p**2)


def beeston_barlow_root1(a, p, U, d):
    """blueice/likelihood.py:693-700 (the root the reference asserts to be <= 0)."""
    return ((-U*p - U + a*p + d*p - np.sqrt(_bb_disc(a, p, U, d))) / (2*p*(p + 1)))


def beeston_barlow_root2(a, p, U, d):
    """blueice/likelihood.py:703-708 (the physical root)."""
    return ((-U*p - U + a*p + d*p + np.sqrt(_bb_disc(a, p, U, d))) / (2*p*(p + 1)))


def adjust_expectations_bb(mus, pmfs, n_model_events, counts, source_i, forgive_zero_u=False):
    """a6: Beeston-Barlow single-source adjustment; returns (mus', pmfs').

    Raises AssertionError exactly where the reference asserts (likelihood.py:649,655).
    forgive_zero_u=True is NOT the reference: in bins where the other sources expect exactly nothing (U_b == 0) the
    first root is 0 analytically, and its floating-point value -- x - sqrt(x^2 (1 +- eps)) -- is negative, zero or
    positive by the last bit of its inputs, so the reference's first assertion fires there on a coin flip.  The switch
    relaxes exactly that assertion (a diagnostic aid; since round 2 the device reproduces the coin and no test uses it)."""
    mus = np.array(mus, dtype=float, copy=True)
    pmfs = np.array(pmfs, dtype=float, copy=True)
    assert pmfs.shape == n_model_events.shape
    per_bin = pmfs.copy()
    for i, (mu, row) in enumerate(zip(mus, per_bin)):
        if i != source_i:
            row *= mu
        else:
            row *= 0.
    u_bins = np.sum(per_bin, axis=0)
    a_bins = n_model_events[source_i]
    with np.errstate(all='ignore'):
        p_cal = mus[source_i] / n_model_events[source_i].sum()
        w_cal = pmfs[source_i] / a_bins * n_model_events[source_i].sum()
        A1 = beeston_barlow_root1(a_bins, w_cal * p_cal, u_bins, counts)
        A2 = beeston_barlow_root2(a_bins, w_cal * p_cal, u_bins, counts)
        if forgive_zero_u:
            assert np.all((A1 <= 0) | ((u_bins == 0) & ~np.isnan(A1)))
        else:
            assert np.all(A1 <= 0)
        A_special = (counts + a_bins) / (1. + p_cal)
        A = np.choose(u_bins == 0, [A2, A_special])
        assert np.all(0 <= A)
        pmfs[source_i] = A * w_cal
        pmfs[source_i] /= pmfs[source_i].sum()
        mus[source_i] = (A * w_cal).sum() * p_cal
    return mus, pmfs


def in_bounds(anchor_z_arrays, z):
    """likelihood.py:345-347: `not minbound <= z <= maxbound` -> -inf (also catches nan)."""
    return all(bool(g[0] <= zi <= g[-1]) for g, zi in zip(anchor_z_arrays, z))


def rates_at(model, z, rate_scale):
    """a4: r_s = mus_interpolator(z)_s * rate_scale_s, rate_scale = rate multiplier * livetime scaling *
    efficiency (blueice/likelihood.py:355,366-393)."""
    mus = interpolate(model['anchor_z'], model['mus'], z)
    return mus * np.asarray(rate_scale, dtype=float)


def loglikelihood(model, counts, z, rate_scale, bb_source=None, allow_negative=None, forgive_zero_u=False):
    """One evaluation of the hot path (a3 + a4 + [a6] + a5) on explicit tensors.

    model: dict(anchor_z=[d arrays], ps=[A.., S, *bins], mus=[A.., S], n_model=None or like ps)
    counts: [*bins];  z: [d];  rate_scale: [S];  bb_source: None or int.
    Mirrors LogLikelihoodBase.__call__ (likelihood.py:318-427) without priors."""
    anchor_z = model['anchor_z']
    z = np.asarray(z, dtype=float)
    if not in_bounds(anchor_z, z):
        return -np.inf
    mus = rates_at(model, z, rate_scale)
    S = len(mus)
    if allow_negative is None or not any(allow_negative):
        if not np.all((mus >= 0) & (mus < np.inf)):
            return -np.inf
    else:
        if (not any(mus < np.inf)) or (np.sum(mus) < 0):
            return -np.inf
        for mu, an in zip(mus, allow_negative):
            if not (0 <= mu) and (not an):
                return -np.inf
    ps = interpolate(anchor_z, model['ps'], z)
    if bb_source is not None:
        n_model = interpolate(anchor_z, model['n_model'], z)
        mus, ps = adjust_expectations_bb(mus, ps, n_model, np.asarray(counts, float), bb_source, forgive_zero_u)
    return float(compute_likelihood(mus, ps.reshape(S, -1), np.asarray(counts, float).ravel()))


def loglikelihood_batch(model, counts, zs, rate_scales, dataset=None, bb_source=None,
                        allow_negative=None):
    """Loop of `loglikelihood` over P points; `counts` is [*bins] or [T, *bins] with
    `dataset[P]` selecting the row (the per-element semantics of the batched entry points)."""
    zs = np.atleast_2d(np.asarray(zs, dtype=float))
    rate_scales = np.atleast_2d(np.asarray(rate_scales, dtype=float))
    out = np.empty(len(zs))
    for i, (z, rs) in enumerate(zip(zs, rate_scales)):
        c = counts if dataset is None else counts[int(dataset[i])]
        try:
            out[i] = loglikelihood(model, c, z, rs, bb_source=bb_source, allow_negative=allow_negative)
        except AssertionError:
            out[i] = np.nan
    return out


def extended_loglikelihood(mu, ps, outlier_likelihood=0.0):
    """blueice/likelihood.py:678-690."""
    mu = np.asarray(mu, dtype=float)
    with np.errstate(all='ignore'):
        p_events = np.nansum(mu[:, np.newaxis] * np.asarray(ps, dtype=float), axis=0)
        if outlier_likelihood != 0:
            p_events[True ^ (p_events > 0)] = outlier_likelihood
        return -mu.sum() + np.sum(np.log(p_events))


def loglikelihood_unbinned(model, z, rate_scale, outlier_likelihood=1e-12, allow_negative=None):
    """One evaluation of UnbinnedLogLikelihood on explicit tensors: model['ps'] is [A.., S, N_events]
    (pdf of every source at every event for every anchor), no counts.  Mirrors
    LogLikelihoodBase.__call__ (likelihood.py:318-427) without priors."""
    anchor_z = model['anchor_z']
    z = np.asarray(z, dtype=float)
    if not in_bounds(anchor_z, z):
        return -np.inf
    mus = rates_at(model, z, rate_scale)
    if allow_negative is None or not any(allow_negative):
        if not np.all((mus >= 0) & (mus < np.inf)):
            return -np.inf
    ps = interpolate(anchor_z, model['ps'], z)
    return float(extended_loglikelihood(mus, ps.reshape(len(mus), -1), outlier_likelihood))
