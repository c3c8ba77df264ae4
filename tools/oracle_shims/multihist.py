"""Minimal stand-in for the third-party `multihist` package (absent from this image, no network).

Build-authored code, NOT reference code.  Used ONLY in the development container by
tests/golden/make_golden.py so that `/root/reference` can be imported to generate golden
vectors; it never travels to (or is imported on) the GPU box and is not part of the product.

Only the surface blueice touches is provided (call sites: blueice/likelihood.py:608-609,
blueice/source.py:229-243,253-254,287-315).  `Histdd.add` is assumed to have
`numpy.histogramdd` edge semantics; golden fixtures therefore place every event strictly
inside a bin and additionally record the binned counts, so the pinned quantities (the
a3-a6 arithmetic of SURVEY.md section 8) do not depend on this stand-in.
"""
from copy import deepcopy

import numpy as np


class Histdd:
    def __init__(self, *data, bins=None, axis_names=None, weights=None):
        self.bin_edges = [np.asarray(b, dtype=float) for b in bins]
        self.axis_names = axis_names
        self.histogram = np.zeros([len(b) - 1 for b in self.bin_edges], dtype=float)
        if len(data):
            self.add(*data, weights=weights)

    @property
    def dimensions(self):
        return len(self.bin_edges)

    def add(self, *data, weights=None):
        sample = np.array([np.asarray(x, dtype=float) for x in data]).T
        if sample.ndim == 1:
            sample = sample.reshape(-1, self.dimensions)
        h, _ = np.histogramdd(sample, bins=self.bin_edges, weights=weights)
        self.histogram = self.histogram + h

    @property
    def n(self):
        return self.histogram.sum()

    def similar_blank_hist(self):
        other = deepcopy(self)
        other.histogram = np.zeros_like(self.histogram)
        return other

    def bin_centers(self, axis=None):
        if axis is None:
            return [0.5 * (e[1:] + e[:-1]) for e in self.bin_edges]
        e = self.bin_edges[axis]
        return 0.5 * (e[1:] + e[:-1])

    def lookup(self, *coords):
        idx = []
        for e, x in zip(self.bin_edges, coords):
            i = np.searchsorted(e, np.asarray(x, dtype=float)) - 1
            idx.append(np.clip(i, 0, len(e) - 2))
        return self.histogram[tuple(idx)]

    def __mul__(self, other):
        out = deepcopy(self)
        out.histogram = self.histogram * other
        return out

    def get_random(self, size=10):
        flat = self.histogram.ravel()
        p = flat / flat.sum()
        picks = np.random.choice(len(flat), size=size, p=p)
        multi = np.unravel_index(picks, self.histogram.shape)
        out = np.zeros((size, self.dimensions))
        for ax, (e, i) in enumerate(zip(self.bin_edges, multi)):
            out[:, ax] = e[i] + np.random.rand(size) * (e[i + 1] - e[i])
        return out
