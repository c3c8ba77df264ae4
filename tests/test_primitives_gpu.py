"""The library's own device-wide primitives (csrc/tu_prim.hip: stable least-significant-digit radix sort of (key, value) pairs, three-
launch scans) against numpy: every key kind, bit ranges, sizes around the 2048-element tiles, negative / infinite / nan doubles,
stability, summation order of the floating-point scan reproducible."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from blueice_amd.device import DeviceContext
    c = DeviceContext(0)
    yield c
    c.close()


def _sort(ctx, kind, keys, vals, b0=0, b1=64):
    from blueice_amd._capi import ptr
    ko, vo = np.empty_like(keys), np.empty_like(vals)
    ctx._check(ctx._lib.bi_selftest_sort(ctx._h, kind, len(keys), ptr(keys), ptr(vals), b0, b1, ptr(ko), ptr(vo)))
    return ko, vo


def _scan(ctx, kind, a, init=0):
    from blueice_amd._capi import ptr
    out = np.empty_like(a)
    ctx._check(ctx._lib.bi_selftest_scan(ctx._h, kind, len(a), ptr(a), int(init), ptr(out)))
    return out


@pytest.mark.parametrize('n', [0, 1, 7, 2047, 2048, 2049, 5000, 100003, 1 << 20])
def test_sorts_equal_numpy_stable_sorts(ctx, n):
    rng = np.random.default_rng(n + 1)
    # unsigned keys with many ties (the planner's (cell, dataset) keys), values = the original positions: stability shows in them
    keys = rng.integers(0, 126, n).astype(np.uint64)
    vals = np.arange(n, dtype=np.int64)
    ko, vo = _sort(ctx, 0, keys, vals, 0, 7)
    order = np.argsort(keys, kind='stable')
    np.testing.assert_array_equal(ko, keys[order])
    np.testing.assert_array_equal(vo, vals[order])
    for bound in (126, 1, 2, 1000, 1024):          # the counting sort: the same stable order
        small = rng.integers(0, bound, n).astype(np.uint64)
        ko, vo = _sort(ctx, 3, small, vals, 0, bound)
        order = np.argsort(small, kind='stable')
        np.testing.assert_array_equal(ko, small[order])
        np.testing.assert_array_equal(vo, vals[order])
    wide = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64)   # all 64 bits
    ko, vo = _sort(ctx, 0, wide, vals)
    order = np.argsort(wide, kind='stable')
    np.testing.assert_array_equal(ko, wide[order])
    np.testing.assert_array_equal(vo, vals[order])
    # a range of bits only (an odd number of passes: 12 bits from bit 5)
    ko, vo = _sort(ctx, 0, wide, vals, 5, 17)
    order = np.argsort((wide >> np.uint64(5)) & np.uint64(0xFFF), kind='stable')
    np.testing.assert_array_equal(ko, wide[order])
    np.testing.assert_array_equal(vo, vals[order])
    # signed keys, int32 values
    sk = rng.integers(-1000, 1000, n).astype(np.int64)
    sv = np.arange(n, dtype=np.int32)
    ko, vo = _sort(ctx, 1, sk, sv)
    order = np.argsort(sk, kind='stable')
    np.testing.assert_array_equal(ko, sk[order])
    np.testing.assert_array_equal(vo, sv[order])
    # doubles: counts (small non-negative integers, ties), and the odd ones -- negative, non-integer, infinite, nan
    dk = rng.poisson(3.0, n).astype(np.float64)
    if n > 20:
        dk[[3, 5, 8, 11, 12, 17]] = [-2.5, np.inf, -np.inf, np.nan, 1e-300, 1e300]     # (-0.0 sorts before +0.0 here -- the bit order,
                                                                                           #  as rocPRIM's did; numpy calls them equal)
    ko, vo = _sort(ctx, 2, dk, sv)
    order = np.argsort(dk, kind='stable')                     # (numpy puts nan last, as the bit order of positive nans does)
    np.testing.assert_array_equal(vo[:n - int(np.isnan(dk).sum())], sv[order][:n - int(np.isnan(dk).sum())])
    np.testing.assert_array_equal(np.isnan(ko), np.isnan(dk[order]))
    finite = ~np.isnan(ko)
    np.testing.assert_array_equal(ko[finite], dk[order][finite])


@pytest.mark.parametrize('n', [0, 1, 255, 2048, 2049, 70001, (1 << 21) + 5])
def test_scans_equal_numpy(ctx, n):
    rng = np.random.default_rng(n + 7)
    a = rng.integers(-5, 50, n).astype(np.int64)
    np.testing.assert_array_equal(_scan(ctx, 1, a), np.cumsum(a))
    np.testing.assert_array_equal(_scan(ctx, 0, a), np.maximum.accumulate(a) if n else a)
    ex = _scan(ctx, 3, a, init=17)
    np.testing.assert_array_equal(ex, 17 + np.concatenate([[0], np.cumsum(a)[:-1]]) if n else a)
    x = rng.random(n) * 3
    got = _scan(ctx, 2, x)
    # (against extended-precision running sums: numpy's own float64 cumsum adds one element after the other and drifts by ~sqrt(n) ulp)
    np.testing.assert_allclose(got, np.cumsum(x.astype(np.longdouble)).astype(np.float64), rtol=1e-13)
    np.testing.assert_array_equal(_scan(ctx, 2, x), got)        # a fixed summation order: the same bits every time
    assert np.all(np.diff(got) >= 0)                             # (the cumulative sums toy generation bisects in are monotonic)
