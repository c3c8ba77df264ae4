"""Sources: the objects that hand the likelihood its templates.

Host-side and one-time (template building is dominated by user simulators and is out of the GPU hot
path), but needed so that the configs the reference's users write -- a model
config dict plus per-source dicts, `default_source_class`, `FixedSampleSource`-style density
estimation -- produce the same tensors here.  Interface mirrored from blueice/source.py:

    Source                   config defaults, `expected_events`                 (:33-189)
    HistogramPdfSource       pdf lookup / simulate / get_pmf_grid from a histogram  (:192-267)
    DensityEstimatingSource  histogram built from a sample of events            (:270-323)
    MonteCarloSource         that sample comes from the source's own simulate() (:326-348)

The reference's on-disk pickle cache (cache_dir, task_dir, hashes) is not reproduced: the
corresponding config keys are accepted and ignored.
"""
import inspect

import numpy as np

from .exceptions import PDFNotComputedException
from .histdd import Histdd
from .utils import combine_dicts, events_to_analysis_dimensions

__all__ = ['Source', 'HistogramPdfSource', 'DensityEstimatingSource', 'MonteCarloSource']


class Source:
    """Base class: bookkeeping of rates.  Subclass and implement `pdf` / `get_pmf_grid` / `simulate`."""

    DEFAULTS = dict(name='unnamed_source', label='Unnamed source', color='black',
                    events_per_day=0, rate_multiplier=1, fraction_in_range=1,
                    delay_pdf_computation=False, dont_hash_settings=[], livetime_days=1)

    def __init__(self, config, *args, **kwargs):
        c = combine_dicts(self.DEFAULTS, config)
        c['dont_hash_settings'] = list(c['dont_hash_settings']) + list(c.pop('extra_dont_hash_settings', []))
        self.name = c.pop('name')
        if hasattr(self, 'events_per_day'):
            raise ValueError("events_per_day defaults should be set via config!")
        self.events_per_day = c['events_per_day']
        self.fraction_in_range = c['fraction_in_range']
        self.pdf_has_been_computed = False
        self.config = c
        if not c['delay_pdf_computation']:
            self.compute_pdf()

    def __repr__(self):
        return "%s[%s]" % (type(self).__name__, self.name)

    def compute_pdf(self):
        """Subclasses build their templates here and call this at the end."""
        if self.pdf_has_been_computed:
            raise RuntimeError("compute_pdf called twice on a source!")
        self.pdf_has_been_computed = True

    @property
    def expected_events(self):
        """Events expected inside the analysis space (blueice/source.py:186-189)."""
        return self.events_per_day * self.config['livetime_days'] * self.fraction_in_range * \
            self.config['rate_multiplier']

    def pdf(self, *args):
        raise NotImplementedError

    def get_pmf_grid(self, *args):
        """-> (pmf per analysis bin, MC events per bin behind it or inf)."""
        raise NotImplementedError

    def simulate(self, n_events):
        raise NotImplementedError


class HistogramPdfSource(Source):
    """PDF given by a histogram: `_pdf_histogram` (density), `_bin_volumes`, `_n_events_histogram`."""

    _pdf_histogram = None
    _bin_volumes = None
    _n_events_histogram = None

    def __init__(self, config, *args, **kwargs):
        config = combine_dicts(dict(pdf_sampling_multiplier=1, pdf_interpolation_method='linear'), config)
        super().__init__(config, *args, **kwargs)

    def build_histogram(self):
        raise NotImplementedError

    def compute_pdf(self):
        self.build_histogram()
        super().compute_pdf()

    def _require_pdf(self, what):
        if not self.pdf_has_been_computed:
            raise PDFNotComputedException("%s: attempt to %s before the PDF was computed" % (self, what))

    def pdf(self, *args):
        self._require_pdf('call the PDF')
        method = self.config['pdf_interpolation_method']
        if method == 'piecewise':
            return self._pdf_histogram.lookup(*args)
        if method == 'linear':
            from scipy.interpolate import RegularGridInterpolator
            if not hasattr(self, '_pdf_interpolator'):
                self._pdf_interpolator = RegularGridInterpolator(self._pdf_histogram.bin_centers(),
                                                                 self._pdf_histogram.histogram)
            clipped = [np.clip(x, c.min(), c.max()) for x, c in zip(args, self._pdf_histogram.bin_centers())]
            return self._pdf_interpolator(np.transpose(clipped))
        raise NotImplementedError("PDF Interpolation method %s not implemented" % method)

    def simulate(self, n_events):
        self._require_pdf('simulate events')
        pts = (self._pdf_histogram * self._bin_volumes).get_random(n_events)
        space = self.config['analysis_space']
        d = np.zeros(n_events, dtype=[('source', int)] + [(name, float) for name, _ in space])
        for i, (name, _) in enumerate(space):
            d[name] = pts[:, i]
        return d

    def get_pmf_grid(self):
        return self._pdf_histogram.histogram * self._bin_volumes, self._n_events_histogram.histogram


class DensityEstimatingSource(HistogramPdfSource):
    """Histogram-density estimate from a sample (`get_events_for_density_estimate`)."""

    def __init__(self, config, *args, **kwargs):
        super().__init__(combine_dicts(dict(n_events_for_pdf=1e6), config), *args, **kwargs)

    def build_histogram(self):
        names, edges = zip(*self.config['analysis_space'])
        counts = Histdd(bins=edges, axis_names=names)
        getter = self.get_events_for_density_estimate
        batches = getter() if inspect.isgeneratorfunction(getter) else [getter()]
        n_total = 0
        for events, n_simulated in batches:
            n_total += n_simulated
            counts.add(*events_to_analysis_dimensions(events, self.config['analysis_space']))
        self.fraction_in_range = counts.n / n_total
        self._bin_volumes = counts.bin_volumes()
        density = counts.similar_blank_hist()
        density.histogram = counts.histogram.astype(float) / counts.n
        density.histogram /= self._bin_volumes
        self._pdf_histogram = density
        self._n_events_histogram = counts
        return counts

    def get_events_for_density_estimate(self):
        """Return (events, number simulated), or yield such pairs in batches."""
        raise NotImplementedError


class MonteCarloSource(DensityEstimatingSource):
    """The density-estimation sample comes from the source's own `simulate`."""

    def __init__(self, config, *args, **kwargs):
        config = combine_dicts(dict(n_events_for_pdf=1e6, pdf_sampling_multiplier=1, pdf_sampling_batch_size=1e6),
                               config)
        super().__init__(config, *args, **kwargs)

    def get_events_for_density_estimate(self):
        n_events = self.config['n_events_for_pdf'] * self.config['pdf_sampling_multiplier']
        batch = min(self.config['pdf_sampling_batch_size'], n_events)
        for _ in range(int(n_events // batch)):
            yield self.simulate(n_events=int(batch)), batch
