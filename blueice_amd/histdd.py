"""A d-dimensional histogram with fixed, possibly non-uniform bin edges.

The reference delegates binning to the third-party `multihist.Histdd`; only the operations the
binned path needs are provided here, with numpy.histogramdd edge semantics (right-most edge
inclusive, out-of-range events dropped).  Call sites being mirrored: blueice/likelihood.py:607-609,
blueice/source.py:229-243,253-254,287-315.
"""
import contextlib
import threading

import numpy as np

__all__ = ['Histdd', 'device_histograms']

# Template building on the device: while `device_histograms(ctx)` is active, Histdd.add bins unweighted batches of at
# least `min_events` events with ctx.histogram_events (bi_histogram_events: the kernel that also bins the data,
# numpy.histogramdd semantics, counts are integers so the result is the same array) instead of numpy.histogramdd,
# which takes about a second per 10^6 three-dimensional events -- times sources x anchor models in prepare().
_engine = None          # (context, min_events) or None
_engine_lock = threading.Lock()      # a device context is not re-entrant; prepare(n_cores > 1) builds models on threads


@contextlib.contextmanager
def device_histograms(ctx, min_events=32768):
    """Route large Histdd.add calls to `ctx` (a DeviceContext) for the duration of the block."""
    global _engine
    previous = _engine
    _engine = (ctx, int(min_events)) if ctx is not None and hasattr(ctx, 'histogram_events') else None
    try:
        yield
    finally:
        _engine = previous


class Histdd:
    def __init__(self, *data, bins, axis_names=None, weights=None):
        self.bin_edges = [np.asarray(e, dtype=float) for e in bins]
        for e in self.bin_edges:
            if e.ndim != 1 or len(e) < 2 or np.any(np.diff(e) <= 0):
                raise ValueError("bin edges must be one-dimensional and strictly increasing")
        self.axis_names = list(axis_names) if axis_names is not None else None
        self.histogram = np.zeros(self.shape, dtype=float)
        if data:
            self.add(*data, weights=weights)

    @property
    def shape(self):
        return tuple(len(e) - 1 for e in self.bin_edges)

    @property
    def dimensions(self):
        return len(self.bin_edges)

    @property
    def n(self):
        """Total content."""
        return self.histogram.sum()

    def add(self, *coords, weights=None):
        """Fill with events given as one coordinate array per axis."""
        if len(coords) != self.dimensions:
            raise ValueError("need %d coordinate arrays, got %d" % (self.dimensions, len(coords)))
        cols = [np.asarray(c, dtype=float).ravel() for c in coords]
        engine = _engine
        if engine is not None and weights is None and len(cols[0]) >= engine[1]:
            with _engine_lock:
                self.histogram += engine[0].histogram_events(self.bin_edges, cols)
            return self
        sample = np.stack(cols, axis=1)
        if len(sample):
            self.histogram += np.histogramdd(sample, bins=self.bin_edges, weights=weights)[0]
        return self

    def similar_blank_hist(self):
        return Histdd(bins=self.bin_edges, axis_names=self.axis_names)

    def bin_centers(self, axis=None):
        mids = [0.5 * (e[:-1] + e[1:]) for e in self.bin_edges]
        return mids if axis is None else mids[axis]

    def bin_volumes(self):
        vol = np.ones(self.shape)
        for ax, e in enumerate(self.bin_edges):
            shape = [1] * self.dimensions
            shape[ax] = -1
            vol = vol * np.diff(e).reshape(shape)
        return vol

    def bin_index(self, *coords):
        """Per-axis bin indices of points, clipped into the histogram range."""
        return tuple(np.clip(np.searchsorted(e, np.asarray(x, dtype=float)) - 1, 0, len(e) - 2)
                     for e, x in zip(self.bin_edges, coords))

    def lookup(self, *coords):
        return self.histogram[self.bin_index(*coords)]

    def __mul__(self, factor):
        out = self.similar_blank_hist()
        out.histogram = self.histogram * factor
        return out

    def get_random(self, size, rng=None):
        """`size` points distributed like the histogram content, uniform within a bin -> [size, d]."""
        rng = np.random if rng is None else rng
        flat = self.histogram.ravel()
        picks = rng.choice(len(flat), size=int(size), p=flat / flat.sum())
        multi = np.unravel_index(picks, self.shape)
        out = np.empty((int(size), self.dimensions))
        for ax, (e, i) in enumerate(zip(self.bin_edges, multi)):
            out[:, ax] = e[i] + rng.random(int(size)) * (e[i + 1] - e[i])
        return out
