"""Toy sources and config builders for the test-suite.

The names (`GaussianSource`, `GaussianMCSource`, `FixedSampleSource`, `conf_for_test`, `make_data`,
`almost_equal`) are the fixtures the reference's tests are written against (blueice/test_helpers.py), kept so
that parity tests read like the reference's own; the implementations are this package's.
"""
import numpy as np
from scipy.stats import norm

from .source import DensityEstimatingSource, MonteCarloSource, Source

__all__ = ['GaussianSourceBase', 'GaussianSource', 'GaussianMCSource', 'FixedSampleSource',
           'BASE_CONFIG', 'BASE_CONV_CONFIG', 'conf_for_test', 'conf_for_reparam_test', 'almost_equal', 'make_data']

_EVENT_DTYPE = [('x', float), ('source', int)]


class GaussianSourceBase(Source):
    """One observable, x ~ Normal(config['mu'], config['sigma'])."""

    def _dist(self):
        return norm(loc=self.config['mu'], scale=self.config['sigma'])

    def simulate(self, n_events):
        events = np.zeros(int(n_events), dtype=_EVENT_DTYPE)
        events['x'] = self._dist().rvs(int(n_events))
        return events


class GaussianSource(GaussianSourceBase):
    """Closed-form pdf.  Two settings act on the rate only: `some_multiplier` (a number) and
    `strlen_multiplier` (a string, the rate scales with its length) -- handy stand-ins for numeric and
    non-numeric shape parameters."""

    def compute_pdf(self):
        factor = self.config.get('some_multiplier', 1) * len(self.config.get('strlen_multiplier', 'x'))
        self.events_per_day = self.events_per_day * factor
        super().compute_pdf()

    def pdf(self, *coords):
        if not self.pdf_has_been_computed:
            raise RuntimeError("Trying to call a PDF that hasn't been computed!")
        return self._dist().pdf(coords[0])


class GaussianMCSource(GaussianSourceBase, MonteCarloSource):
    """Same generator; the pdf is a histogram of its own Monte Carlo."""


class FixedSampleSource(DensityEstimatingSource):
    """Histogram density of the events given in config['data']; rate x len(strlen_multiplier)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.events_per_day = self.events_per_day * len(self.config.get('strlen_multiplier', 'x'))

    def get_events_for_density_estimate(self):
        sample = self.config['data']
        return sample, len(sample)


def _base_config():
    return {
        'analysis_space': [['x', np.linspace(-10, 10, 100)]],
        'default_source_class': GaussianSource,
        'sources': [{'name': 's0', 'events_per_day': 1000.}],
        'events_per_day': 1000.,
        'mu': 0, 'sigma': 1,
        'some_multiplier': 1, 'strlen_multiplier': 'q',
        'n_events_for_pdf': int(1e6),
        'force_pdf_recalculation': True,
    }


BASE_CONFIG = _base_config()


def conf_for_test(n_sources=1, mc=False, **overrides):
    """Model config with `n_sources` identical sources named s0, s1, ...; `mc` switches to GaussianMCSource."""
    conf = _base_config()
    conf['sources'] = [{'name': 's%d' % i} for i in range(n_sources)]
    if mc:
        conf['default_source_class'] = GaussianMCSource
    conf.update(overrides)
    return conf


def conf_for_reparam_test(n_source=1, mc=False, **overrides):
    """Three sources op0, op1, op2 and two extra settings np0, np1 for LogLikelihoodReParam to convert from."""
    conf = conf_for_test(n_source, mc, **overrides)
    conf['sources'] = [{'name': 'op%d' % i} for i in range(3)]
    conf['np0'] = conf['np1'] = 1
    return conf


# conversion: op0 = np0^2, op1 = np1^2, op2 = np0 np1 (relative to their values at np0 = np1 = 1)
BASE_CONV_CONFIG = dict(
    np0=(np.linspace(1e-12, 10, 2), None, None),
    np1=(np.linspace(1e-12, 10, 2), None, None),
    op0_rate_multiplier=dict(params=['np0'], func=lambda a: a ** 2),
    op1_rate_multiplier=dict(params=['np1'], func=lambda b: b ** 2),
    op2_rate_multiplier=dict(params=['np0', 'np1'], func=lambda a, b: a * b),
)


def almost_equal(a, b, fraction=1e-6):
    return abs((a - b) / a) <= fraction


def make_data(instructions):
    """Events from a recipe: each dict gives n_events and constant values for some of the fields
    (source, x, y).  -> (record array, number of events)."""
    counts = [int(step['n_events']) for step in instructions]
    events = np.zeros(sum(counts), dtype=[('source', int), ('x', float), ('y', float)])
    edges = np.concatenate([[0], np.cumsum(counts)]).astype(int)
    for step, lo, hi in zip(instructions, edges[:-1], edges[1:]):
        for field, value in step.items():
            if field != 'n_events':
                events[field][lo:hi] = value
    return events, int(edges[-1])
