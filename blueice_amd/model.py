"""Model: a list of sources sharing one config (host side; mirrors blueice/model.py:8-112)."""
import numpy as np

from .utils import combine_dicts, events_to_analysis_dimensions

__all__ = ['Model']


class Model:
    def __init__(self, config, **kwargs):
        defaults = dict(livetime_days=1, data_dirs=1,
                        nohash_settings=['data_dirs', 'pdf_sampling_batch_size', 'force_recalculation'])
        self.config = combine_dicts(defaults, config, kwargs, deep_copy=True)
        if 'rate_multiplier' in self.config:
            raise ValueError("Don't put a setting named rate_multiplier in the model config please...")
        self.sources = []
        for spec in self.config['sources']:
            cls = spec.get('class', self.config.get('default_source_class'))
            conf = combine_dicts(self.config, spec, exclude=['sources', 'default_source_class', 'class'])
            # `<name>_rate_multiplier` model settings scale that one source (model.py:38-41)
            name = conf.get('name', '')
            own = conf.get('%s_rate_multiplier' % name, 1)
            conf = {k: v for k, v in conf.items() if not k.endswith('_rate_multiplier')}
            conf['rate_multiplier'] = own
            self.sources.append(cls(conf))
        del self.config['sources']

    # -- lookup ----------------------------------------------------------------------------
    def get_source_i(self, source_id):
        if isinstance(source_id, (int, float, np.integer)):
            return int(source_id)
        for i, s in enumerate(self.sources):
            if source_id in s.name:
                return i
        raise ValueError("Unknown source %s" % source_id)

    def get_source(self, source_id):
        return self.sources[self.get_source_i(source_id)]

    # -- data ------------------------------------------------------------------------------
    def to_analysis_dimensions(self, d):
        return events_to_analysis_dimensions(d, self.config['analysis_space'])

    def range_cut(self, d):
        keep = np.ones(len(d), dtype=bool)
        for name, edges in self.config['analysis_space']:
            keep &= (d[name] >= edges[0]) & (d[name] <= edges[-1])
        return d[keep]

    def simulate(self, rate_multipliers=None, livetime_days=None):
        """Toy dataset: Poisson-fluctuated numbers of events from every source (model.py:69-91)."""
        rate_multipliers = rate_multipliers or {}
        parts = []
        for i, s in enumerate(self.sources):
            mu = s.expected_events * rate_multipliers.get(s.name, 1) / s.fraction_in_range
            if livetime_days is not None:
                mu *= livetime_days / self.config['livetime_days']
            ev = s.simulate(np.random.poisson(mu))
            ev['source'] = i
            parts.append(ev)
        return self.range_cut(np.concatenate(parts))

    # -- what the likelihood consumes ------------------------------------------------------
    def expected_events(self, s=None):
        if s is None:
            return np.array([src.expected_events for src in self.sources])
        return s.expected_events

    def pmf_grids(self):
        """-> (pmf [S, *bins], MC events per bin [S, *bins])  (model.py:101-104)."""
        grids = [s.get_pmf_grid() for s in self.sources]
        return np.stack([g[0] for g in grids]), np.stack([g[1] for g in grids])

    def score_events(self, d):
        return np.vstack([s.pdf(*self.to_analysis_dimensions(d)) for s in self.sources])
