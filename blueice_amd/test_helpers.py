"""Analytic toy sources and config builders used by the test-suite (the counterparts of the
fixtures in the reference's blueice/test_helpers.py:13-126; same names so tests read alike)."""
from copy import deepcopy

import numpy as np
from scipy import stats

from .source import DensityEstimatingSource, MonteCarloSource, Source
from .utils import combine_dicts

__all__ = ['GaussianSourceBase', 'GaussianSource', 'GaussianMCSource', 'FixedSampleSource',
           'BASE_CONFIG', 'conf_for_test', 'almost_equal', 'make_data']


class GaussianSourceBase(Source):
    """1-d source that can draw events from N(mu, sigma)."""

    def simulate(self, n_events):
        d = np.zeros(n_events, dtype=[('x', float), ('source', int)])
        d['x'] = stats.norm(self.config['mu'], self.config['sigma']).rvs(n_events)
        return d


class GaussianSource(GaussianSourceBase):
    """Analytic Gaussian pdf; rate responds to `some_multiplier` and len(`strlen_multiplier`)."""

    def compute_pdf(self):
        self.events_per_day *= self.config.get('some_multiplier', 1)
        self.events_per_day *= len(self.config.get('strlen_multiplier', 'x'))
        super().compute_pdf()

    def pdf(self, *args):
        if not self.pdf_has_been_computed:
            raise RuntimeError("Trying to call a PDF that hasn't been computed!")
        return stats.norm(self.config['mu'], self.config['sigma']).pdf(args[0])


class GaussianMCSource(GaussianSourceBase, MonteCarloSource):
    """Same events, pdf estimated from its own Monte Carlo."""


class FixedSampleSource(DensityEstimatingSource):
    """Density estimated from the fixed sample in config['data']."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.events_per_day *= len(self.config.get('strlen_multiplier', 'x'))

    def get_events_for_density_estimate(self):
        return self.config['data'], len(self.config['data'])


BASE_CONFIG = dict(
    sources=[{'name': 's0', 'events_per_day': 1000.}],
    mu=0, sigma=1, strlen_multiplier='q', some_multiplier=1,
    events_per_day=1000., n_events_for_pdf=int(1e6),
    default_source_class=GaussianSource,
    force_pdf_recalculation=True,
    analysis_space=[['x', np.linspace(-10, 10, 100)]],
)


def conf_for_test(n_sources=1, mc=False, **kwargs):
    conf = deepcopy(BASE_CONFIG)
    conf['sources'] = [{'name': 's%d' % i} for i in range(n_sources)]
    if mc:
        conf['default_source_class'] = GaussianMCSource
    return combine_dicts(conf, kwargs)


def almost_equal(a, b, fraction=1e-6):
    return abs((a - b) / a) <= fraction


def make_data(instructions):
    """[dict(n_events=24, x=0.5), dict(n_events=56, x=1.5)] -> (record array, total events)."""
    n_tot = sum(ins['n_events'] for ins in instructions)
    d = np.zeros(n_tot, dtype=[('source', int), ('x', float), ('y', float)])
    start = 0
    for ins in instructions:
        stop = start + ins['n_events']
        for k, v in ins.items():
            if k != 'n_events':
                d[k][start:stop] = v
        start = stop
    return d, n_tot
