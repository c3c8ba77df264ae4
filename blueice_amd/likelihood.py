"""Binned log-likelihood with rate and shape parameters, evaluated on the GPU.

Same surface as the reference's LogLikelihoodBase / BinnedLogLikelihood (blueice/likelihood.py:53-675):

    lf = BinnedLogLikelihood(pdf_base_config, likelihood_config=None, **overrides)
    lf.add_rate_parameter(name, log_prior) / lf.add_shape_parameter(name, anchors, log_prior, base_value)
    lf.prepare(); lf.set_data(d); lf(**params) -> float          [full_output=True -> (ll, mus, ps)]
    lf.bestfit_scipy(...), lf.make_objective(...)                 (inference.py, attached below)

What differs is where the work happens.  `prepare()` streams the anchor models' PMF grids into HBM,
`set_data()` uploads the binned counts, and every call does only the scalar bookkeeping of
likelihood.py:328-415 on the host (parameter validation, bounds, priors, rate multipliers) before ONE
fused device call performs the morph (likelihood.py:355-357), the optional Beeston-Barlow adjustment
(:618-660) and the Poisson reduction (:662-675).  Batched entry points (`eval_points`, `eval_toys`)
expose what the reference can only do as Python loops (inference.py:49-50,424-432).
"""
from collections import OrderedDict
from copy import deepcopy
from functools import wraps

import numpy as np
from scipy import stats

from . import _capi
from .device import DeviceContext
from .exceptions import DeviceError, InvalidParameter, InvalidParameterSpecification, NotPreparedException
from .histdd import Histdd
from .model import Model
from .pdf_morphers import MORPHERS
from .utils import combine_dicts, is_numeric

__all__ = ['LogLikelihoodBase', 'BinnedLogLikelihood', 'UnbinnedLogLikelihood', 'LogLikelihoodSum',
           'LogLikelihoodReParam', 'LogAncillaryLikelihood']

_BB_FLAGS = _capi.ST_BB_ROOT1 | _capi.ST_BB_NEG


def _needs_preparation(method):
    @wraps(method)
    def guarded(self, *args, **kwargs):
        if not self.is_prepared:
            if len(self.shape_parameters):
                raise NotPreparedException("%s requires you to first prepare the likelihood function using prepare()"
                                           % method.__name__)
            self.prepare()      # nothing to morph: preparation is trivial
        return method(self, *args, **kwargs)
    return guarded


def _needs_data(method):
    @wraps(method)
    def guarded(self, *args, **kwargs):
        if not self.is_data_set:
            raise NotPreparedException("%s requires you to first set the data using set_data()" % method.__name__)
        return method(self, *args, **kwargs)
    return guarded


class LogLikelihoodBase:
    """Parameter registry, anchor-model construction and call plumbing shared by likelihoods."""

    def __init__(self, pdf_base_config, likelihood_config=None, **kwargs):
        self.pdf_base_config = combine_dicts(pdf_base_config, kwargs, deep_copy=True)
        self.config = {} if likelihood_config is None else likelihood_config
        self.config.setdefault('morpher', 'GridInterpolator')
        # every source morphs over the shape parameters IT depends on only (likelihood.py:76,152-171); unbinned likelihoods
        self.source_wise_interpolation = self.pdf_base_config.get('source_wise_interpolation', False)

        self.base_model = Model(self.pdf_base_config)
        sources = self.base_model.sources
        self.source_name_list = [s.name for s in sources]
        self.source_allowed_negative = [s.config.get('allow_negative', False) for s in sources]
        self.source_apply_efficiency = np.array([s.config.get('apply_efficiency', False) for s in sources])
        self.source_efficiency_names = np.array([s.config.get('efficiency_name', 'efficiency') for s in sources])

        self.rate_parameters = OrderedDict()     # source name -> log prior (or None)
        self.shape_parameters = OrderedDict()    # setting name -> (anchors {z: setting}, log prior, base z)
        self.anchor_models = OrderedDict()       # z tuple -> Model
        self.anchor_sources = OrderedDict()      # source-wise interpolation: source name -> own z tuple -> Source
        self.is_prepared = False
        self.is_data_set = False
        self._has_non_numeric = False
        self.morpher = None

    # -- parameters ------------------------------------------------------------------------
    def add_rate_parameter(self, source_name, log_prior=None):
        """`<source_name>_rate_multiplier` multiplies that source's expected events."""
        self.rate_parameters[source_name] = log_prior

    def add_shape_parameter(self, setting_name, anchors, log_prior=None, base_value=None):
        """Vary `setting_name` over `anchors`: a sequence of numeric settings, or a dict z -> setting
        for non-numeric settings (then `base_value` names the z of the base model's setting)."""
        numeric = is_numeric(self.pdf_base_config.get(setting_name))
        if not isinstance(anchors, dict):
            if not numeric:
                raise InvalidParameterSpecification("When specifying anchors only by setting values, "
                                                    "base setting must have a numerical default.")
            anchors = {z: z for z in anchors}
        if numeric and base_value is not None:
            raise InvalidParameterSpecification("For numeric settings, base_value is an unnecessary argument.")
        if not numeric:
            if base_value is None:
                raise InvalidParameterSpecification("For non-numeric settings, you must specify what number will "
                                                    "represent the default value (the base model setting)")
            self._has_non_numeric = True
        self.shape_parameters[setting_name] = (anchors, log_prior, base_value)

    def add_rate_uncertainty(self, source_name, fractional_uncertainty):
        self.add_rate_parameter(source_name, log_prior=stats.norm(1, fractional_uncertainty).logpdf)

    def add_shape_uncertainty(self, setting_name, fractional_uncertainty, anchor_zs=(-2, -1, 0, 1, 2), base_value=None):
        self.add_shape_parameter(setting_name, anchor_zs, base_value=base_value)
        anchors, _, base_value = self.shape_parameters[setting_name]
        prior = stats.norm(base_value, base_value * fractional_uncertainty).logpdf
        self.shape_parameters[setting_name] = (anchors, prior, base_value)

    def get_bounds(self, parameter_name=None):
        if parameter_name is None:
            return [self.get_bounds(p) for p in self.shape_parameters]
        if parameter_name in self.shape_parameters:
            zs = list(self.shape_parameters[parameter_name][0].keys())
            return min(zs), max(zs)
        if parameter_name.endswith('_rate_multiplier'):
            for name, neg in zip(self.source_name_list, self.source_allowed_negative):
                if parameter_name.startswith(name) and neg:
                    return float('-inf'), float('inf')
            return 0, float('inf')
        raise InvalidParameter("Non-existing parameter %s" % parameter_name)

    def _kwargs_to_settings(self, **kwargs):
        """-> (rate multipliers [S], {shape setting: z}) with defaults filled in."""
        for k in kwargs:
            known = k in self.shape_parameters or (k.endswith('_rate_multiplier')
                                                   and k[:-len('_rate_multiplier')] in self.source_name_list)
            if not known:
                raise InvalidParameter("%s is not a known shape or rate parameter!" % k)
        settings = {}
        for name, (_, _, base_value) in self.shape_parameters.items():
            z = kwargs.get(name)
            if z is None:
                base = self.pdf_base_config.get(name)
                z = base if is_numeric(base) else base_value
            if not is_numeric(z):
                raise ValueError("Arguments to likelihood function must be numeric, not %s" % type(z))
            settings[name] = z
        multipliers = [kwargs.get(s + '_rate_multiplier', 1) for s in self.source_name_list]
        return multipliers, settings

    # -- source-wise interpolation -----------------------------------------------------------
    @property
    def source_shape_parameters(self):
        """source name -> the shape parameters that source responds to: all of them except the settings its config
        lists under dont_hash_settings (an efficiency setting the source applies always counts); sources that respond
        to none are left out (likelihood.py:113-130)."""
        out = OrderedDict()
        for name, source, use_eff, eff_name in zip(self.source_name_list, self.base_model.sources,
                                                   self.source_apply_efficiency, self.source_efficiency_names):
            ignored = set(source.config['dont_hash_settings'])
            if use_eff:
                ignored.discard(eff_name)
            own = OrderedDict((k, v) for k, v in self.shape_parameters.items() if k not in ignored)
            if own:
                out[name] = own
        return out

    def _get_shape_indices(self, source_name):
        """Positions, among all shape parameters, of the ones `source_name` responds to."""
        own = self.source_shape_parameters[source_name]
        return [i for i, k in enumerate(self.shape_parameters) if k in own]

    def _get_model_anchor(self, anchor, source_name):
        """A source's own anchor -> the anchor of the model that holds it: None where the source does not care."""
        full = [None] * len(self.shape_parameters)
        for z, i in zip(anchor, self._get_shape_indices(source_name)):
            full[i] = z
        return tuple(full)

    def _prepare_source_wise(self, build_all):
        """Models only at the anchors some source needs (a setting no source of the model reads at that anchor stays at
        its base value), then per full-grid anchor a view that takes every source from the model built at ITS
        projection of the anchor.  The device tensor stays the full Cartesian grid: interpolating a source over an
        axis it does not respond to returns its value unchanged (the weights along that axis sum to one), so the
        result is the reference's per-source interpolation (likelihood.py:152-171,210-240,534-563) and the kernels
        do not change; the reduced set of models is what the user's sources see built."""
        own_params = self.source_shape_parameters
        own_morphers = OrderedDict((name, MORPHERS[self.config['morpher']](self.config.get('morpher_config', {}), sp))
                                   for name, sp in own_params.items())
        wanted = []
        for name, morpher in own_morphers.items():
            for anchor in morpher.get_anchor_points(bounds=None):
                zs = self._get_model_anchor(tuple(anchor), name)
                if zs not in wanted:
                    wanted.append(zs)
        built = dict(zip(wanted, build_all(wanted)))
        self.source_morphers = own_morphers
        self.anchor_sources = OrderedDict()
        for name, morpher in own_morphers.items():
            i = self.source_name_list.index(name)
            self.anchor_sources[name] = OrderedDict(
                (tuple(anchor), built[self._get_model_anchor(tuple(anchor), name)].sources[i])
                for anchor in morpher.get_anchor_points(bounds=None))
        for zs in (tuple(z) for z in self.morpher.get_anchor_points(bounds=self.get_bounds())):
            sources = []
            for i, name in enumerate(self.source_name_list):
                if name in own_params:
                    sources.append(self.anchor_sources[name][tuple(zs[j] for j in self._get_shape_indices(name))])
                else:
                    sources.append(self.base_model.sources[i])
            self.anchor_models[zs] = _SourceWiseModel(self.base_model, sources)

    # -- anchor models ---------------------------------------------------------------------
    def prepare(self, n_cores=1, ipp_client=None):
        """Compute the model at every anchor point.  n_cores > 1 builds the anchor models on a pool of THREADS
        (numpy-heavy sources release the interpreter lock while they histogram / sample); the reference's
        process-pool / ipyparallel farms with their on-disk cache, likelihood.py:184-208, are out of scope --
        `ipp_client` is accepted and ignored."""
        self.anchor_models = OrderedDict()
        if len(self.shape_parameters):
            self.morpher = MORPHERS[self.config['morpher']](self.config.get('morpher_config', {}),
                                                            self.shape_parameters)

            def build(zs):
                conf = deepcopy(self.pdf_base_config)
                for z, (name, (anchors, _, _)) in zip(zs, self.shape_parameters.items()):
                    if z is not None:                      # None: no source built here reads this setting
                        conf[name] = anchors[z]
                return Model(conf)

            def build_all(points):
                with self._template_building():
                    if n_cores and n_cores > 1 and len(points) > 1:
                        from concurrent.futures import ThreadPoolExecutor
                        with ThreadPoolExecutor(max_workers=int(n_cores)) as pool:
                            return list(pool.map(build, points))
                    return [build(zs) for zs in points]

            if self.source_wise_interpolation:
                self._prepare_source_wise(build_all)
            else:
                points = [tuple(zs) for zs in self.morpher.get_anchor_points(bounds=self.get_bounds())]
                for zs, model in zip(points, build_all(points)):   # anchor order, whatever order the pool finished in
                    self.anchor_models[zs] = model
        self.is_data_set = False
        self.is_prepared = True

    def _template_building(self):
        """Context manager around the construction of the anchor models (a hook: the device likelihoods bin the
        sources' Monte Carlo samples on the GPU meanwhile)."""
        from contextlib import nullcontext
        return nullcontext()

    @_needs_preparation
    def set_data(self, d):
        self._data = d
        self.is_data_set = True
        for per_source in self.anchor_sources.values():        # scores cached for the previous dataset (_SourceWiseModel)
            for source in per_source.values():
                source.__dict__.pop('_blueice_amd_score', None)
        for source in self.base_model.sources:
            source.__dict__.pop('_blueice_amd_score', None)

    def _compute_single_model(self, **kwargs):
        _, settings = self._kwargs_to_settings(**kwargs)
        return Model(combine_dicts(self.pdf_base_config, settings, deep_copy=True))

    # -- host half of one evaluation -------------------------------------------------------
    def _host_terms(self, livetime_days, kwargs, morph=True, mus_of=None):
        """Everything of likelihood.py:328-393 that is scalar bookkeeping.
        -> (prior_sum, z vector, rate_scale [S]) or (None, None, None) when z is out of bounds.
        morph=False is the compute_pdf branch (likelihood.py:331-335): the model is built AT the settings, so there is
        no anchor-box test and the shape priors are not added (the reference adds them only on the interpolating
        branch, :341-350); rate priors, live-time and efficiency scaling apply either way.
        mus_of() -> expected events [S] before rate multipliers; only called for the zero-live-time assertion."""
        multipliers, settings = self._kwargs_to_settings(**kwargs)
        prior = 0
        zs = []
        for name, (_, log_prior, _) in self.shape_parameters.items():
            z = settings[name]
            zs.append(z)
            if not morph:
                continue
            lo, hi = self.get_bounds(name)
            if not lo <= z <= hi:
                return None, None, None         # cannot extrapolate: -inf (likelihood.py:345-347)
            if log_prior is not None:
                prior += log_prior(z)
        scale = np.array(multipliers, dtype=float)
        for mult, name in zip(multipliers, self.source_name_list):
            log_prior = self.rate_parameters.get(name)
            if log_prior is not None:
                prior += log_prior(mult)
        if livetime_days is not None:
            if 'livetime_days' not in self.pdf_base_config:
                raise ValueError("Cannot scale live-time, base value absent")
            base = self.pdf_base_config['livetime_days']
            if base == 0:
                if livetime_days != 0:
                    raise ValueError("Cannot scale from 0 to non-0 livetime")
                # likelihood.py:380: with no live time every source must expect exactly nothing
                mus = mus_of() if mus_of is not None else self.ctx.interpolate('mus', np.asarray(zs, dtype=float))
                assert np.all(np.asarray(mus, dtype=float) * scale == 0), "Got non-0 mus with 0 livetime?!"
            else:
                scale = scale * (livetime_days / base)
        if True in self.source_apply_efficiency:
            effs = [settings.get(name, 1) for use, name in
                    zip(self.source_apply_efficiency, self.source_efficiency_names) if use]
            scale[self.source_apply_efficiency] *= np.array(effs)
        return prior, np.asarray(zs, dtype=float), scale


class _SourceWiseModel:
    """What a full-grid anchor looks like under source-wise interpolation: the base model's analysis space with every
    source taken from the model built at that source's own anchor.  Scores are cached per source object, so a source
    shared by many grid anchors is evaluated once per dataset."""

    def __init__(self, base_model, sources):
        self.config = base_model.config
        self.sources = sources
        self._base = base_model

    def expected_events(self, s=None):
        if s is None:
            return np.array([src.expected_events for src in self.sources])
        return s.expected_events

    def score_events(self, d):
        coords = self._base.to_analysis_dimensions(d)
        rows = []
        for src in self.sources:
            cached = getattr(src, '_blueice_amd_score', None)
            if cached is None or cached[0] is not d:
                cached = (d, src.pdf(*coords))
                src._blueice_amd_score = cached
            rows.append(cached[1])
        return np.vstack(rows)


def _prior_of(log_prior, values):
    """log_prior over an array of parameter values: one vectorised call when the callable takes arrays (scipy's
    frozen distributions do), the reference's one call per value otherwise."""
    try:
        out = np.asarray(log_prior(values), dtype=float)
        if out.shape == values.shape:
            return out
    except Exception:
        pass
    return np.array([log_prior(v) for v in values], dtype=float)


class DeviceLogLikelihood(LogLikelihoodBase):
    """What the binned and the unbinned likelihood share: one DeviceContext holding the anchor tensor, and
    evaluation = host bookkeeping + one fused device call."""

    model_statistical_uncertainty_handling = None

    def __init__(self, pdf_base_config, likelihood_config=None, **kwargs):
        super().__init__(pdf_base_config, likelihood_config, **kwargs)
        self.ps = self.n_model_events = None
        self.ctx = None
        self.bin_shape = ()

    def _bb_source_index(self):
        return -1

    def _template_building(self):
        """While the anchor models are built, density-estimating sources histogram their samples on the device
        (`histdd.device_histograms`; blueice/source.py:287-299 does it with multihist on the host -- at C2 size that is
        500 source templates x ~1 s).  likelihood_config['device_histograms'] = False keeps numpy.histogramdd."""
        from contextlib import nullcontext
        from .histdd import device_histograms
        if not self.config.get('device_histograms', True):
            return nullcontext()
        if self.ctx is None:
            self.ctx = DeviceContext(self.config.get('device'))
        return device_histograms(self.ctx, self.config.get('device_histograms_min_events', 32768))

    def _stream_models(self, pmf_of, n_bins):
        """Fill the device context from the anchor models (or the base model when nothing morphs).
        pmf_of(model) -> (rows [S, n_bins], MC counts [S, n_bins] or None)."""
        S = len(self.source_name_list)
        bb = self._bb_source_index()
        if self.ctx is None:
            self.ctx = DeviceContext(self.config.get('device'))
        if len(self.shape_parameters):
            self.morpher.stream_to_device(self.ctx, self.anchor_models, S, n_bins, bb_source=bb, rows_of=pmf_of)
        else:
            rows, n_mc = pmf_of(self.base_model)
            self.ctx.begin_model([], S, n_bins, bb_source=bb)
            self.ctx.set_anchor(0, rows, self.base_model.expected_events(), n_mc[bb] if bb >= 0 else None)
            self.ctx.end_model()
        self.ctx.set_allow_negative([1 if x else 0 for x in self.source_allowed_negative])

    # the morpher closures of the reference, served from the same device context
    def mus_interpolator(self, zs):
        return self.ctx.interpolate('mus', zs)

    def ps_interpolator(self, zs):
        return self.ctx.interpolate('ps', zs).reshape((len(self.source_name_list),) + tuple(self.bin_shape))

    # hooks of the two concrete likelihoods
    def _rows_of(self, model):
        """-> (template rows [S, *bin_shape] of `model`, MC counts or None)."""
        raise NotImplementedError

    def _attach_data(self, ctx):
        """Give a scratch context the same data this likelihood holds."""
        raise NotImplementedError

    # -- evaluation ------------------------------------------------------------------------
    def _interpret(self, ll, status, mus_hint=None):
        if status & _capi.ST_INTERNAL:
            raise DeviceError("the device gave up waiting for a partial sum (in-launch reduction): GPU fault")
        if status & _capi.ST_UNPHYSICAL:
            if self.config.get('unphysical_behaviour') == 'error':
                raise ValueError("Unphysical rates: %s" % str(mus_hint))
            return -float('inf')
        if status & _capi.ST_BB_ROOT1:
            raise AssertionError("Beeston-Barlow: first root is not negative everywhere")
        if status & _capi.ST_BB_NEG:
            raise AssertionError("Beeston-Barlow: negative adjusted expectation")
        return ll

    @_needs_data
    def __call__(self, livetime_days=None, compute_pdf=False, full_output=False, **kwargs):
        if compute_pdf and len(self.shape_parameters):
            if self._has_non_numeric:
                raise NotImplementedError("compute_pdf only works for numerical values")
            return self._call_with_fresh_pdf(livetime_days, full_output, kwargs)
        prior, zs, scale = self._host_terms(livetime_days, kwargs)
        if prior is None:
            return -float('inf')
        if full_output:
            ll, mus, ps, st = self.ctx.eval_full(zs, scale)
            ll = self._interpret(ll, st, mus)
            if ll == -float('inf') and st:
                return ll
            return prior + ll, mus, ps.reshape((len(mus),) + tuple(self.bin_shape))
        ll, st = self.ctx.eval_one(zs, scale)
        return self._finish_call(prior, zs, scale, ll, st)

    def _finish_call(self, prior, zs, scale, ll, st):
        if st:
            hint = self.ctx.interpolate('mus', zs) * scale if st & _capi.ST_UNPHYSICAL else None
            ll0 = self._interpret(ll, st, hint)
            return ll0 if ll0 == -float('inf') else prior + ll0
        return prior + ll

    # The plain call in two halves, for LogLikelihoodSum: begin() on every term, then end() on every term, so the
    # device work of the terms (one context each) overlaps.
    @_needs_data
    def _call_begin(self, livetime_days=None, **kwargs):
        prior, zs, scale = self._host_terms(livetime_days, kwargs)
        if prior is not None:
            self.ctx.eval_begin(zs if len(zs) else None, scale)
        return prior, zs, scale

    def _call_end(self, token):
        prior, zs, scale = token
        if prior is None:
            return -float('inf')
        ll, st = self.ctx.eval_end()
        return self._finish_call(prior, zs, scale, ll, st)

    def _call_with_fresh_pdf(self, livetime_days, full_output, kwargs):
        """compute_pdf=True: build the model AT the requested settings instead of morphing
        (likelihood.py:331-335,611-616) and evaluate it through a scratch d=0 device model.  As in the reference
        this works outside the anchor box too and leaves the shape priors out."""
        model = self._compute_single_model(**kwargs)
        prior, zs, scale = self._host_terms(livetime_days, kwargs, morph=False, mus_of=model.expected_events)
        ps, n_mc = self._rows_of(model)
        bb = self._bb_source_index()
        scratch = DeviceContext(self.ctx.device)
        try:
            scratch.begin_model([], len(ps), int(np.prod(self.bin_shape, dtype=np.int64)), bb_source=bb)
            scratch.set_anchor(0, ps, model.expected_events(), n_mc[bb] if bb >= 0 else None)
            scratch.end_model()
            scratch.set_allow_negative([1 if x else 0 for x in self.source_allowed_negative])
            self._attach_data(scratch)
            if full_output:
                ll, mus, ps_out, st = scratch.eval_full(None, scale)
                ll = self._interpret(ll, st, mus)
                if ll == -float('inf') and st:
                    return ll
                return prior + ll, mus, ps_out.reshape(ps.shape)
            ll, st = scratch.eval(None, scale[None, :])
            ll, st = float(ll[0]), int(st[0])
            if st:
                ll0 = self._interpret(ll, st, model.expected_events() * scale)
                return ll0 if ll0 == -float('inf') else prior + ll0
            return prior + ll
        finally:
            scratch.close()

    # -- batched entry points (no counterpart in the reference) --------------------------------
    def _batch_terms(self, points, livetime_days, want_unit=False):
        """-> (z [P, d], scale [P, S], prior [P]) of a dict of parameter arrays; want_unit=True adds `unit` [P, S], the
        scale per unit rate multiplier (live-time and efficiency factors: d scale / d multiplier)."""
        names = list(points.keys())
        cols = [np.atleast_1d(np.asarray(points[n], dtype=float)) for n in names]
        P = max((len(c) for c in cols), default=1)
        cols = [c if len(c) == P else np.broadcast_to(c, (P,)) for c in cols]
        known = self.__dict__.get('_batch_names')          # (validated names and defaults: rebuilt when the parameters change)
        # (the defaults are base values: part of the key, so that a changed base value is seen -- the scalar call reads
        # them afresh every time, and batched and scalar evaluations of one point must agree)
        key = (tuple(self.shape_parameters), tuple(self.source_name_list),
               tuple((self.pdf_base_config.get(n), sp[2]) for n, sp in self.shape_parameters.items()))
        if known is None or known[0] != key:
            known = self._batch_names = (key, set(), self._kwargs_to_settings()[1])
        for k in names:
            if k not in known[1]:
                self._kwargs_to_settings(**{k: 0.0})       # name validation only
                known[1].add(k)
        defaults = known[2]
        z = np.empty((P, len(self.shape_parameters)))
        prior = np.zeros(P)
        for i, (name, (_, log_prior, _)) in enumerate(self.shape_parameters.items()):
            z[:, i] = cols[names.index(name)] if name in names else defaults[name]
            if log_prior is not None:
                prior += _prior_of(log_prior, z[:, i])
        scale = np.ones((P, len(self.source_name_list)))
        unit = np.ones_like(scale) if want_unit else None
        for s, name in enumerate(self.source_name_list):
            key = name + '_rate_multiplier'
            if key in names:
                scale[:, s] = cols[names.index(key)]
            log_prior = self.rate_parameters.get(name)
            if log_prior is not None:
                prior += _prior_of(log_prior, scale[:, s])
        if livetime_days is not None:
            if 'livetime_days' not in self.pdf_base_config:
                raise ValueError("Cannot scale live-time, base value absent")
            base = self.pdf_base_config['livetime_days']
            if base == 0:
                if livetime_days != 0:
                    raise ValueError("Cannot scale from 0 to non-0 livetime")
                lo_hi = [self.get_bounds(n) for n in self.shape_parameters]
                for zi, sc in zip(z, scale):        # likelihood.py:380, per point; points outside the box return -inf first
                    if all(lo <= v <= hi for v, (lo, hi) in zip(zi, lo_hi)):
                        assert np.all(self.ctx.interpolate('mus', zi) * sc == 0), "Got non-0 mus with 0 livetime?!"
            else:
                scale = scale * (livetime_days / base)
                if want_unit:
                    unit = unit * (livetime_days / base)
        if True in self.source_apply_efficiency:
            for s in np.flatnonzero(self.source_apply_efficiency):
                en = self.source_efficiency_names[s]
                eff = cols[names.index(en)] if en in names else defaults.get(en)
                if eff is not None:
                    scale[:, s] *= eff
                    if want_unit:
                        unit[:, s] *= eff
        return (z, scale, prior, unit) if want_unit else (z, scale, prior)

    @_needs_data
    def eval_points(self, points, livetime_days=None, dataset=None):
        """Evaluate many parameter points in one device call.

        points: dict parameter name -> array [P] (scalars broadcast; absent parameters take their
        defaults) -- the batched form of `[lf(**kw) for kw in ...]` used by likelihood scans,
        `best_anchor` and profile grids.  Returns ll [P]; out-of-bounds / unphysical points are -inf
        (or raise, under unphysical_behaviour='error'), exactly as the scalar call."""
        z, scale, prior = self._batch_terms(points, livetime_days)
        ll, st = self.ctx.eval(z if z.shape[1] else None, scale, dataset)
        if np.any(st & _capi.ST_INTERNAL):
            raise DeviceError("the device gave up waiting for a partial sum (in-launch reduction): GPU fault")
        bad = st & _capi.ST_UNPHYSICAL
        if np.any(bad) and self.config.get('unphysical_behaviour') == 'error':
            raise ValueError("Unphysical rates at %d of %d points" % (int(np.count_nonzero(bad)), len(st)))
        if np.any(st & _BB_FLAGS):
            raise AssertionError("Beeston-Barlow assertion at %d points" % int(np.count_nonzero(st & _BB_FLAGS)))
        out = ll + prior
        out[(st & (_capi.ST_OUT_OF_BOUNDS | _capi.ST_UNPHYSICAL)) != 0] = -np.inf
        return out

    @staticmethod
    def _prior_slope(log_prior, x):
        if log_prior is None:
            return 0.0
        h = 1e-6 * max(1.0, abs(x))
        return (log_prior(x + h) - log_prior(x - h)) / (2 * h)

    @property
    def supports_gradient(self):
        """bi_eval_grad covers binned likelihoods, with and without Beeston-Barlow (there the chain rule runs through the
        per-bin root of likelihood.py:693-712), and the extended unbinned likelihood (d log lambda_e = d lambda_e / lambda_e,
        events on the outlier clamp contribute no slope; likelihood.py:678-690), up to 1 + d + S = 16 gradient columns
        (d <= 7 with Beeston-Barlow)."""
        n = 1 + len(self.shape_parameters) + len(self.source_name_list)
        if self.model_statistical_uncertainty_handling is not None and len(self.shape_parameters) > 7:
            return False
        return n <= 16

    @_needs_data
    def value_and_gradient(self, livetime_days=None, **kwargs):
        """-> (ll, OrderedDict parameter name -> d ll / d parameter) for every registered rate and shape
        parameter, from ONE pass over the templates (`bi_eval_grad`).  Inside a grid cell ll is smooth in
        the shape parameters; exactly on an anchor the slope of the cell the point is assigned to is
        returned.  Prior terms are differentiated numerically on the host (they are Python callables)."""
        prior, zs, scale = self._host_terms(livetime_days, kwargs)
        grads = OrderedDict()
        names = ['%s_rate_multiplier' % s for s in self.rate_parameters] + list(self.shape_parameters)
        if prior is None:
            return -float('inf'), OrderedDict((n, float('nan')) for n in names)
        multipliers, settings = self._kwargs_to_settings(**kwargs)
        ll, gz, gs, st = self.ctx.eval_grad(zs if len(zs) else None, scale[None, :])
        ll = self._interpret(float(ll[0]), int(st[0]))
        gz, gs = gz[0], gs[0]
        mult = np.array(multipliers, dtype=float)
        with np.errstate(all='ignore'):
            per_mult = np.where(mult != 0, scale / np.where(mult != 0, mult, 1.0), 0.0)
        if np.any(mult == 0):          # d scale / d multiplier does not depend on the multiplier itself
            _, _, unit = self._host_terms(livetime_days, {k: v for k, v in kwargs.items()
                                                          if not k.endswith('_rate_multiplier')})
            per_mult = np.where(mult != 0, per_mult, unit)
        for s, name in enumerate(self.source_name_list):
            if name in self.rate_parameters:
                grads['%s_rate_multiplier' % name] = gs[s] * per_mult[s] + \
                    self._prior_slope(self.rate_parameters[name], multipliers[s])
        for i, (name, (_, log_prior, _)) in enumerate(self.shape_parameters.items()):
            g = gz[i] + self._prior_slope(log_prior, settings[name])
            # a shape parameter that doubles as the efficiency of some sources also scales their rates
            for s in np.flatnonzero(self.source_apply_efficiency):
                if self.source_efficiency_names[s] == name and settings[name] != 0:
                    g += gs[s] * scale[s] / settings[name]
            grads[name] = g
        return prior + ll, grads

    @_needs_data
    def values_and_gradients(self, points, livetime_days=None, dataset=None, bb_assert='raise'):
        """The batched form of `value_and_gradient`: points = dict parameter name -> array [P] (as `eval_points`) ->
        (ll [P], OrderedDict parameter name -> d ll / d parameter [P]) for every registered rate and shape parameter,
        from ONE device call (`bi_eval_grad` over all P points).  Points outside the anchor box or with unphysical
        rates give -inf and nan slopes.  What the batched profile-fit engine (blueice_amd.profile) advances P
        minimisations with.  bb_assert: points at which one of the reference's Beeston-Barlow assertions would fire
        (likelihood.py:649,655) raise AssertionError, as the scalar call does -- or, with 'nan', come back as nan so that a
        line search can step around them."""
        z, scale, prior, unit = self._batch_terms(points, livetime_days, want_unit=True)
        P = len(z)
        ll, gz, gs, st = self.ctx.eval_grad(z if z.shape[1] else None, scale, dataset)
        if np.any(st & _capi.ST_INTERNAL):
            raise DeviceError("the device gave up waiting for a partial sum (in-launch reduction): GPU fault")
        if np.any(st & _BB_FLAGS):
            if bb_assert != 'nan':
                raise AssertionError("Beeston-Barlow assertion at %d points" % int(np.count_nonzero(st & _BB_FLAGS)))
            ll = np.where(st & _BB_FLAGS, np.nan, ll)
        bad = (st & (_capi.ST_OUT_OF_BOUNDS | _capi.ST_UNPHYSICAL)) != 0
        if np.any(st & _capi.ST_UNPHYSICAL) and self.config.get('unphysical_behaviour') == 'error':
            raise ValueError("Unphysical rates at %d of %d points" % (int(np.count_nonzero(st & _capi.ST_UNPHYSICAL)), P))
        out = ll + prior
        out[bad] = -np.inf

        def slope(log_prior, x):               # priors are Python callables: central differences, vectorised
            if log_prior is None:
                return 0.0
            h = 1e-6 * np.maximum(1.0, np.abs(x))
            return (_prior_of(log_prior, x + h) - _prior_of(log_prior, x - h)) / (2 * h)

        grads = OrderedDict()
        for s, name in enumerate(self.source_name_list):
            if name in self.rate_parameters:
                mult = np.broadcast_to(np.asarray(points.get(name + '_rate_multiplier', 1.0), dtype=float), (P,))
                grads['%s_rate_multiplier' % name] = gs[:, s] * unit[:, s] + slope(self.rate_parameters[name], mult)
        for i, (name, (_, log_prior, _)) in enumerate(self.shape_parameters.items()):
            g = gz[:, i] + slope(log_prior, z[:, i])
            # a shape parameter that doubles as the efficiency of some sources also scales their rates
            for s in np.flatnonzero(self.source_apply_efficiency):
                if self.source_efficiency_names[s] == name:
                    with np.errstate(all='ignore'):
                        g = g + np.where(z[:, i] != 0, gs[:, s] * scale[:, s] / np.where(z[:, i] != 0, z[:, i], 1.0), 0.0)
            grads[name] = g
        for g in grads.values():
            g[bad] = np.nan
        return out, grads

    # -- toy-MC ---------------------------------------------------------------------------------


class BinnedLogLikelihood(DeviceLogLikelihood):
    """Poisson likelihood over the bins of the analysis space, morph + reduce fused on the GPU."""

    def __init__(self, pdf_base_config, likelihood_config=None, **kwargs):
        super().__init__(pdf_base_config, likelihood_config, **kwargs)
        pdf_base_config['pdf_interpolation_method'] = 'piecewise'
        self.model_statistical_uncertainty_handling = self.config.get('model_statistical_uncertainty_handling')
        self._lazy_nm_interpolator = None
        self._binned = None

    # -- lifecycle -------------------------------------------------------------------------
    def _bb_source_index(self):
        if self.model_statistical_uncertainty_handling is None:
            return -1
        if self.model_statistical_uncertainty_handling != 'bb_single':
            raise NotImplementedError("model_statistical_uncertainty_handling=%r" %
                                      self.model_statistical_uncertainty_handling)
        src = self.config.get('bb_single_source')
        if src is None:
            raise ValueError("You need to specify bb_single_source to use bb_single_source expectation adjustment")
        return self.base_model.get_source_i(src)

    def prepare(self, *args, **kwargs):
        super().prepare(*args, **kwargs)
        self.ps, self.n_model_events = self.base_model.pmf_grids()
        if len(self.shape_parameters) and self.source_wise_interpolation:
            raise NotImplementedError("Source-wise interpolation not implemented for binned likelihoods")   # likelihood.py:590-591
        self.bin_shape = self.ps.shape[1:]
        self._stream_models(lambda m: m.pmf_grids(), int(np.prod(self.bin_shape, dtype=np.int64)))
        self.ctx.set_analysis_space([edges for _, edges in self.base_model.config['analysis_space']])
        self._lazy_nm_interpolator = None
        self._binned = None

    def _rows_of(self, model):
        return model.pmf_grids()

    def _attach_data(self, ctx):
        ctx.upload_counts(self.data_events_per_bin.histogram)

    def n_model_events_interpolator(self, zs):
        if self.model_statistical_uncertainty_handling is None or not len(self.shape_parameters):
            return None
        if self._lazy_nm_interpolator is None:      # full [S, *bins] tensor only if somebody asks for it
            self._lazy_nm_interpolator = self.morpher.make_interpolator(
                f=lambda m: m.pmf_grids()[1], extra_dims=list(self.ps.shape), anchor_models=self.anchor_models)
        return self._lazy_nm_interpolator(zs)

    @_needs_preparation
    def set_data(self, d):
        """Bin the events of `d` in the analysis space ON THE DEVICE (numpy.histogramdd semantics, as the
        reference's Histdd.add, likelihood.py:608-609); only the event coordinates cross PCIe."""
        LogLikelihoodBase.set_data(self, d)
        self.ctx.upload_events(*self.base_model.to_analysis_dimensions(d))
        self._binned = None

    @property
    def data_events_per_bin(self):
        """The binned data as a histogram object (`.histogram` = counts of dataset 0), fetched from the device
        on first use -- the attribute the reference fills in set_data."""
        if self._binned is None:
            names, edges = zip(*self.base_model.config['analysis_space'])
            self._binned = Histdd(bins=edges, axis_names=names)
            self._binned.histogram = self.ctx.download_counts(0).reshape(self.bin_shape)
        return self._binned

    @_needs_preparation
    def set_binned_data(self, counts):
        """Upload already-binned counts: [*bins] or [T, *bins] for T toy datasets.
        Dataset 0 is what plain `lf(**params)` evaluates."""
        counts = np.asarray(counts, dtype=float)
        if counts.shape[-len(self.bin_shape):] != tuple(self.bin_shape):
            raise ValueError("counts must end in the analysis-space shape %s" % (tuple(self.bin_shape),))
        self._data = None
        self.ctx.upload_counts(counts)
        self._binned = None
        self.is_data_set = True

    @_needs_preparation
    def simulate_toys(self, n_toys, seed=0, livetime_days=None, **kwargs):
        """Draw `n_toys` binned toy datasets ON THE DEVICE at the given parameter values and make them the
        likelihood's data (dataset 0 is what plain `lf(**params)` sees; `eval_toys` evaluates them all).
        Per bin n ~ Poisson(mu_b): the distribution that `base_model.simulate()` + `set_data()` produces
        (model.py:69-91, likelihood.py:603-609), without events or host transfers."""
        prior, zs, scale = self._host_terms(livetime_days, kwargs)
        if prior is None:
            raise ValueError("cannot simulate outside the anchor box")
        self.ctx.generate_toys(zs, scale, n_toys, seed)
        if self.model_statistical_uncertainty_handling is not None:
            self.ctx.counts_to_dense()             # Beeston-Barlow reads n in every bin: the toys as a dense array too
        self._data = None
        self._binned = None
        self.is_data_set = True

    # -- analytic gradient (one device pass; the reference differentiates numerically) ----------
    @_needs_data
    def eval_toys(self, livetime_days=None, t0=0, t1=None, **kwargs):
        """One parameter point against every uploaded dataset (see `set_binned_data`): ll [T]."""
        prior, zs, scale = self._host_terms(livetime_days, kwargs)
        T = (self.ctx.T if t1 is None else t1) - t0
        if prior is None:
            return np.full(T, -np.inf)
        if self.model_statistical_uncertainty_handling is not None:
            ll, st = self.ctx.eval(np.tile(zs, (T, 1)) if len(zs) else None, np.tile(scale, (T, 1)),
                                   np.arange(t0, t0 + T))
            return np.array([prior + self._interpret(a, int(b)) for a, b in zip(ll, st)])
        ll, st = self.ctx.eval_datasets(zs, scale, t0, t0 + T)
        if st & _capi.ST_UNPHYSICAL and self.config.get('unphysical_behaviour') == 'error':
            raise ValueError("Unphysical rates")
        return ll + prior if not st else ll


    @_needs_data
    def eval_toys_points(self, points, livetime_days=None, t0=0, t1=None):
        """Many parameter points against every uploaded / simulated dataset in ONE device call: ll [P, T].

        points: dict parameter name -> array [P] (scalars broadcast; absent parameters take their defaults), as `eval_points`.
        The toy-MC double loop -- every simulated dataset (`base_model.simulate()` + `set_data`, blueice/model.py:69-91) evaluated
        at every hypothesis (the loops of blueice/inference.py:392-443) -- as `bi_eval_datasets_points`: the hypotheses of a
        grid cell share the pass over its templates, four hypotheses the pass over the datasets' lists.  Rows of points outside
        the anchor box or with unphysical rates are -inf (or raise, under unphysical_behaviour='error'), as the scalar call."""
        z, scale, prior = self._batch_terms(points, livetime_days)
        T = (self.ctx.T if t1 is None else t1) - t0
        if self.model_statistical_uncertainty_handling is not None:      # Beeston-Barlow: mu depends on the data -- dataset by point
            return np.stack([self.eval_toys(livetime_days, t0, t0 + T, **{k: float(np.broadcast_to(v, (len(z),))[i]) for k, v in points.items()})
                             for i in range(len(z))])
        ll, st = self.ctx.eval_datasets_points(z if z.shape[1] else None, scale, t0, t0 + T)
        bad = st & _capi.ST_UNPHYSICAL
        if np.any(bad) and self.config.get('unphysical_behaviour') == 'error':
            raise ValueError("Unphysical rates at %d of %d points" % (int(np.count_nonzero(bad)), len(st)))
        out = ll + np.asarray(prior, dtype=float).reshape(-1, 1)
        out[(st & (_capi.ST_OUT_OF_BOUNDS | _capi.ST_UNPHYSICAL)) != 0] = -np.inf
        return out


class UnbinnedLogLikelihood(DeviceLogLikelihood):
    """Extended unbinned likelihood, -sum_s mu_s + sum_events log(sum_s mu_s p_s(x_e)), on the same device
    path (reference: blueice/likelihood.py:528-573, extended_loglikelihood :678-690).  `set_data` scores the
    events at every anchor model on the host (`Model.score_events`, as the reference does, :557-560) and
    streams the [anchor][source][event] pdf tensor to HBM; every call is then one fused morph + reduce.
    Events whose density is not positive get config['outlier_likelihood'] (default 1e-12)."""

    def __init__(self, pdf_base_config, likelihood_config=None, **kwargs):
        super().__init__(pdf_base_config, likelihood_config, **kwargs)
        self.outlier_likelihood = self.config.get('outlier_likelihood', 1e-12)
        self._templates = None            # DeviceContext holding the sources' density histograms, or False: not applicable

    def prepare(self, *args, **kwargs):
        super().prepare(*args, **kwargs)
        if self._templates:
            self._templates[0].close()
        self._templates = None

    def _rows_of(self, model):
        return model.score_events(self._data), None

    def _attach_data(self, ctx):
        ctx.set_unbinned(self.outlier_likelihood)

    # -- set_data on the device, when every source's pdf is a histogram -------------------------------------------
    def _histogram_templates(self):
        """-> (templates context, method, grid) when every source of every anchor model takes its pdf from a histogram
        over the model's analysis space through HistogramPdfSource.pdf itself, all with the same interpolation method;
        None otherwise (analytic pdfs, overridden pdf(), mixed methods): then the events are scored on the host."""
        from .source import HistogramPdfSource
        if self._templates is False or not self.config.get('device_scoring', True) or not hasattr(DeviceContext, 'score_events'):
            return None
        if self._templates is None:
            self._templates = False
            models = list(self.anchor_models.values()) if len(self.shape_parameters) else [self.base_model]
            sources = {id(s): s for m in models for s in m.sources}.values()
            edges = [np.asarray(e, dtype=float) for _, e in self.base_model.config['analysis_space']]
            methods = set()
            for s in sources:
                if not isinstance(s, HistogramPdfSource) or type(s).pdf is not HistogramPdfSource.pdf \
                        or not s.pdf_has_been_computed or s._pdf_histogram is None:
                    return None
                he = s._pdf_histogram.bin_edges
                if len(he) != len(edges) or any(len(a) != len(b) or np.any(a != b) for a, b in zip(he, edges)):
                    return None
                methods.add(s.config['pdf_interpolation_method'])
            if len(methods) != 1 or methods - {'linear', 'piecewise'}:
                return None
            method = methods.pop()
            if method == 'linear' and any(len(e) < 3 for e in edges):
                return None                      # one bin on an axis: scipy refuses such a grid, let the host say so
            tp = DeviceContext(self.config.get('device'))
            S = len(self.source_name_list)
            n_bins = int(np.prod([len(e) - 1 for e in edges], dtype=np.int64))
            densities = lambda m: (np.stack([s._pdf_histogram.histogram.ravel() for s in m.sources]), None)
            if len(self.shape_parameters):
                self.morpher.stream_to_device(tp, self.anchor_models, S, n_bins, rows_of=densities)
            else:
                tp.begin_model([], S, n_bins)
                tp.set_anchor(0, densities(self.base_model)[0], self.base_model.expected_events())
                tp.end_model()
            tp.set_allow_negative([1 if x else 0 for x in self.source_allowed_negative])
            grid = edges if method == 'piecewise' else [0.5 * (e[:-1] + e[1:]) for e in edges]
            self._templates = (tp, method, grid)
        return self._templates

    @_needs_preparation
    def set_data(self, d):
        """Score the events at every anchor model (likelihood.py:531-563).  Sources whose pdf is a histogram are
        evaluated ON THE DEVICE -- their density histograms are uploaded once, then every set_data sends the event
        coordinates only and one kernel fills the [anchor][source][event] tensor in HBM (`bi_score_events`); other
        sources are scored on the host, anchor by anchor, and the tensor is streamed up."""
        LogLikelihoodBase.set_data(self, d)
        self.bin_shape = (len(d),)
        if not len(self.shape_parameters):
            self.ps = self.base_model.score_events(d)
        coords = [np.asarray(c, dtype=float) for c in self.base_model.to_analysis_dimensions(d)]
        tpl = self._histogram_templates() if all(np.all(np.isfinite(c)) for c in coords) else None
        if tpl:
            tp, method, grid = tpl
            if method == 'linear':               # constant density in the outer half of the boundary bins (source.py:232-241)
                coords = [np.clip(c, g[0], g[-1]) for c, g in zip(coords, grid)]
            if self.ctx is None:
                self.ctx = DeviceContext(self.config.get('device'))
            tp.score_events(self.ctx, method, grid, coords, self.outlier_likelihood)
            return
        self._stream_models(self._rows_of, len(d))
        self.ctx.set_unbinned(self.outlier_likelihood)


    @_needs_preparation
    def simulate_toy(self, seed=0, livetime_days=None, **kwargs):
        """Draw an event-level toy dataset ON THE DEVICE at the given parameter values and make it the likelihood's data:
        what `d = lf.base_model.simulate(...)` + `lf.set_data(d)` do on the host (model.py:69-91, source.py:248-264,
        likelihood.py:531-563) -- per source N_s ~ Poisson(mu_s) events, each a bin drawn with probability density x
        volume of the (morphed) histogram pdf and a uniform position inside it, then scored at every anchor model --
        with nothing but the call crossing PCIe.  Needs sources whose pdf is a histogram (see `set_data`); raises
        NotImplementedError otherwise.  -> events per source; `simulated_events()` fetches the events themselves."""
        tpl = self._histogram_templates()
        if not tpl:
            raise NotImplementedError("device-side event simulation needs sources whose pdf is a histogram over the analysis space")
        prior, zs, scale = self._host_terms(livetime_days, kwargs)
        if prior is None:
            raise ValueError("cannot simulate outside the anchor box")
        tp, method, _ = tpl
        if self.ctx is None:
            self.ctx = DeviceContext(self.config.get('device'))
        edges = [np.asarray(e, dtype=float) for _, e in self.base_model.config['analysis_space']]
        per_source = tp.simulate_events(self.ctx, method, edges, zs, scale, seed, self.outlier_likelihood)
        self._data = None
        self.bin_shape = (int(per_source.sum()),)
        self.is_data_set = True
        return per_source

    def simulated_events(self):
        """The events of the last `simulate_toy` as a record array with the analysis dimensions and a 'source' field, as
        `Model.simulate` returns them."""
        coords, source = self.ctx.download_events()
        names = [n for n, _ in self.base_model.config['analysis_space']]
        d = np.zeros(coords.shape[1], dtype=[(n, float) for n in names] + [('source', int)])
        for n, c in zip(names, coords):
            d[n] = c
        d['source'] = source
        return d


class LogLikelihoodSum:
    """Weighted sum of likelihoods sharing (some) parameters, with the likelihood interface the inference
    helpers need (reference: blueice/likelihood.py:867-955).  A host-side combinator: every term is its own
    device context and is evaluated with the parameters it knows."""

    def __init__(self, likelihood_list, likelihood_weights=None):
        self.likelihood_list = list(likelihood_list)
        self.likelihood_weights = list(likelihood_weights) if likelihood_weights is not None \
            else [1] * len(self.likelihood_list)
        self.rate_parameters, self.shape_parameters = dict(), dict()
        self.pdf_base_config = {}
        self.source_list = []
        self.likelihood_parameters = []
        for ll in self.likelihood_list:
            self.rate_parameters.update(ll.rate_parameters)
            self.shape_parameters.update(ll.shape_parameters)
            names = ['%s_rate_multiplier' % r for r in ll.rate_parameters] + list(ll.shape_parameters)
            for key in list(ll.rate_parameters) + list(ll.shape_parameters):
                if ll.pdf_base_config.get(key) is not None:
                    self.pdf_base_config[key] = ll.pdf_base_config[key]
            self.likelihood_parameters.append(names)

    def __call__(self, compute_pdf=False, livetime_days=None, **kwargs):
        terms = list(zip(self.likelihood_list, self.likelihood_parameters, self.likelihood_weights))
        lts = [livetime_days[i] if isinstance(livetime_days, list) else livetime_days for i in range(len(terms))]
        kws = [{k: v for k, v in kwargs.items() if k in names} for _, names, _ in terms]
        total = 0.
        if not compute_pdf and all(hasattr(ll, '_call_begin') for ll, _, _ in terms) and \
                len({id(getattr(ll, 'ctx', None)) for ll, _, _ in terms}) == len(terms):
            # every term is a device likelihood with its own context: launch them all, then collect them all
            tokens = []
            try:
                for (ll, _, _), lt, kw in zip(terms, lts, kws):
                    tokens.append(ll._call_begin(livetime_days=lt, **kw))
            finally:                                  # whatever was launched is collected, also on an error
                results = []
                for (ll, _, _), token in zip(terms, tokens):
                    try:
                        results.append(ll._call_end(token))
                    except BaseException as err:      # keep collecting; re-raised below
                        results.append(err)
            for r in results:
                if isinstance(r, BaseException):
                    raise r
            for (_, _, weight), r in zip(terms, results):
                total += weight * r
            return total
        for (ll, _, weight), lt, kw in zip(terms, lts, kws):
            total += weight * ll(compute_pdf=compute_pdf, livetime_days=lt, **kw)
        return total

    # -- batched / gradient forms (extensions, as on the single likelihoods) -------------------
    def eval_points(self, points, livetime_days=None):
        """Weighted sum of the terms' `eval_points` over the same dict of parameter arrays; every term sees the
        parameters it knows (a term that knows none of them contributes its constant)."""
        total = 0.
        for i, (ll, names, weight) in enumerate(zip(self.likelihood_list, self.likelihood_parameters, self.likelihood_weights)):
            lt = livetime_days[i] if isinstance(livetime_days, list) else livetime_days
            mine = {k: v for k, v in points.items() if k in names}
            if hasattr(ll, 'eval_points'):
                total = total + weight * ll.eval_points(mine, livetime_days=lt)
                continue
            # a term that only knows scalar calls (an analytic constraint, LogAncillaryLikelihood): point by point
            cols = {k: np.atleast_1d(np.asarray(v, dtype=float)) for k, v in mine.items()}
            P = max([len(c) for c in cols.values()] + [1])
            vals = np.array([ll(**{k: float(c[j] if len(c) == P else c[0]) for k, c in cols.items()}) for j in range(P)])
            total = total + weight * (vals if cols else vals[0])
        return total

    @property
    def supports_gradient(self):
        """True when every term can return an analytic gradient (binned terms without Beeston-Barlow);
        `bestfit_scipy(use_gradient=True)` falls back to numerical differences otherwise."""
        return all(getattr(ll, 'supports_gradient', False) for ll in self.likelihood_list)

    def value_and_gradient(self, livetime_days=None, **kwargs):
        """-> (ll, OrderedDict name -> d ll / d parameter): the weighted sum of the terms' values and gradients
        (one device pass per term); lets `bestfit_scipy(use_gradient=True)` work on a sum."""
        if not self.supports_gradient:
            raise NotImplementedError("a term of this sum has no analytic gradient (unbinned or Beeston-Barlow)")
        total, grads = 0., OrderedDict()
        for i, (ll, names, weight) in enumerate(zip(self.likelihood_list, self.likelihood_parameters, self.likelihood_weights)):
            lt = livetime_days[i] if isinstance(livetime_days, list) else livetime_days
            v, g = ll.value_and_gradient(livetime_days=lt, **{k: x for k, x in kwargs.items() if k in names})
            total += weight * v
            for name, slope in g.items():
                grads[name] = grads.get(name, 0.) + weight * slope
        return total, grads

    def values_and_gradients(self, points, livetime_days=None, **options):
        """The batched form, as on the single likelihoods: points = dict name -> array [P] -> (ll [P], OrderedDict name ->
        slopes [P]), the weighted sums of the terms' (one device call per term).  What the batched profile-fit engine
        iterates with on a sum of likelihoods."""
        if not self.supports_gradient:
            raise NotImplementedError("a term of this sum has no analytic gradient (unbinned)")
        total, grads = 0., OrderedDict()
        for i, (ll, names, weight) in enumerate(zip(self.likelihood_list, self.likelihood_parameters, self.likelihood_weights)):
            lt = livetime_days[i] if isinstance(livetime_days, list) else livetime_days
            v, g = ll.values_and_gradients({k: x for k, x in points.items() if k in names}, livetime_days=lt, **options)
            total = total + weight * v
            for name, slope in g.items():
                grads[name] = grads.get(name, 0.) + weight * slope
        return total, grads

    def split_results(self, result_dict):
        return [{k: v for k, v in result_dict.items() if k in names} for names in self.likelihood_parameters]

    def get_bounds(self, parameter_name=None):
        if parameter_name is None:
            return [self.get_bounds(p) for p in self.shape_parameters]
        if parameter_name in self.shape_parameters:
            b = np.array([ll.get_bounds(parameter_name) for ll in self.likelihood_list
                          if parameter_name in ll.shape_parameters])
            lo, hi = np.max(b[:, 0]), np.min(b[:, 1])
            if hi <= lo:
                raise InvalidParameterSpecification("lower bound %s higher than upper bound!" % parameter_name)
            return lo, hi
        if parameter_name.endswith('_rate_multiplier'):
            return 0, float('inf')
        raise InvalidParameter("Non-existing parameter %s" % parameter_name)


class LogLikelihoodReParam:
    """A likelihood seen through other parameters (reference: blueice/likelihood.py:715-864).  Host-side only: new
    parameters are turned into rate multipliers of the wrapped likelihood before every call, so the device path is
    the wrapped likelihood's.

    conv_config maps
      `<new parameter>`               -> (anchor values, log prior, base value): registers the new parameter (its
                                         bounds are the extremes of the anchor values);
      `<source>_rate_multiplier`      -> dict(params=[new parameters], func=f): the multiplier handed to the wrapped
                                         likelihood is f(*values) / f(*base values), base values from the model config.
    """

    def __init__(self, likelihood, conv_config):
        self._inner = likelihood
        self.conv_config = conv_config
        self.pdf_base_config = likelihood.pdf_base_config
        self.check_conv_config()

    @staticmethod
    def _is_rate(key):
        return key.endswith('_rate_multiplier')

    def check_conv_config(self):
        """The parameters the conversions read are exactly the new parameters declared, and each has a base value."""
        declared = {k for k in self.conv_config if not self._is_rate(k)}
        used = {p for v in self.conv_config.values() if isinstance(v, dict) for p in v['params']}
        assert declared == used, "New parameters are not consistent, double check conv_config..."
        config = self._inner.base_model.config
        missing = [p for p in self.conv_config if not self._is_rate(p) and not config.get(p, False)]
        assert not missing, "%s are missing in the config" % ', '.join(missing)

    # -- what the inference helpers look at ---------------------------------------------------------
    @property
    def rate_parameters(self):
        """The wrapped likelihood's rate parameters that are not computed from new parameters."""
        return OrderedDict((k, v) for k, v in self._inner.rate_parameters.items()
                           if k + '_rate_multiplier' not in self.conv_config)

    @property
    def shape_parameters(self):
        out = OrderedDict(self._inner.shape_parameters)
        for k, v in self.conv_config.items():
            if not self._is_rate(k):
                out[k] = ({z: z for z in v[0]}, v[1], v[2])
        return out

    @property
    def base_model(self):
        """A copy of the wrapped base model whose simulate() understands the new parameters."""
        model = deepcopy(self._inner.base_model)
        model.simulate = self._simulate
        return model

    def set_data(self, d):
        self._inner.set_data(d)

    def get_bounds(self, parameter_name=None):
        if parameter_name is None:
            return [self.get_bounds(p) for p in self.shape_parameters]
        inner = self._inner
        if parameter_name in inner.shape_parameters or parameter_name in inner.rate_parameters \
                or self._is_rate(parameter_name):
            return inner.get_bounds(parameter_name)
        zs = list(self.shape_parameters[parameter_name][0].keys())
        return min(zs), max(zs)

    # -- conversion -----------------------------------------------------------------------------------
    def _parameter_converter(self, with_suffix=True, **kwargs):
        """New-parameter kwargs -> kwargs of the wrapped likelihood.  with_suffix=False: rate multipliers go in and
        come out keyed by source name (the form Model.simulate takes)."""
        if not with_suffix:
            kwargs = {(k + '_rate_multiplier' if k in self._inner.rate_parameters else k): v for k, v in kwargs.items()}
        out, consumed = OrderedDict(), set()
        for key, spec in self.conv_config.items():
            if not self._is_rate(key):
                continue
            base = [self.pdf_base_config.get(p) for p in spec['params']]
            here = [kwargs.get(p, b) for p, b in zip(spec['params'], base)]
            out[key] = spec['func'](*here) / spec['func'](*base)
            consumed.update(spec['params'])
        for k, v in kwargs.items():
            if k not in consumed:
                out[k] = v
        if not with_suffix:
            out = OrderedDict((k.split('_rate_multiplier')[0], v) for k, v in out.items())
        return out

    def __call__(self, compute_pdf=False, livetime_days=None, **kwargs):
        return self._inner(compute_pdf=compute_pdf, livetime_days=livetime_days,
                           **deepcopy(self._parameter_converter(**kwargs)))

    def _simulate(self, kwargs=None, livetime_days=None):
        converted = self._parameter_converter(with_suffix=False, **(kwargs or {}))
        multipliers = {k: v for k, v in converted.items() if k in self._inner.rate_parameters}
        return self._inner.base_model.simulate(rate_multipliers=multipliers, livetime_days=livetime_days)


class LogAncillaryLikelihood:
    """An analytic constraint term with the likelihood interface, for use inside a LogLikelihoodSum (reference:
    blueice/likelihood.py:958-1001).  `func(OrderedDict parameter -> value, **func_kwargs)` returns the log
    likelihood; parameters not given in a call take their value from `config`."""

    def __init__(self, func, parameter_list, config=None, func_kwargs=None):
        self.func = func
        self.func_kwargs = {} if func_kwargs is None else func_kwargs
        self.pdf_base_config = {} if config is None else config
        self.rate_parameters = dict()
        self.source_list = []
        self.shape_parameters = OrderedDict((name, (None, None, None)) for name in parameter_list)

    def get_bounds(self, parameter_name=None):
        if parameter_name is None:
            return [self.get_bounds(p) for p in self.shape_parameters]
        if parameter_name in self.shape_parameters:
            return -np.inf, np.inf                      # the other terms of a sum may be more restrictive
        raise InvalidParameter("Non-existing parameter %s" % parameter_name)

    def __call__(self, **kwargs):
        values = OrderedDict((name, self.pdf_base_config[name]) for name in self.shape_parameters)
        values.update(kwargs)
        return self.func(values, **self.func_kwargs)


# inference helpers double as methods, as in the reference (likelihood.py:1004-1007)
from . import inference  # noqa: E402

for _name in inference.__all__:
    for _cls in (LogLikelihoodBase, LogLikelihoodSum, LogAncillaryLikelihood, LogLikelihoodReParam):
        setattr(_cls, _name, getattr(inference, _name))
