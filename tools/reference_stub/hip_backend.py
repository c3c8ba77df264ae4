"""blueice/hip_backend.py -- the binding a blueice maintainer would add to put libblueice_hip.so under the reference's own
BinnedLogLikelihood (as a real file).

Build-authored code (not part of the reference): it keeps ALL of blueice -- configuration, sources, Model, the morpher's
anchor grid, data binning, priors, exceptions -- and hands the three hot steps to the C ABI of include/blueice_hip.h:

    prepare()   BinnedLogLikelihood.prepare builds the anchor models (blueice/likelihood.py:147-264,586-601); their
                pmf grids / expected events / Monte Carlo counts go to the library one anchor at a time in the order of
                GridInterpolator.make_interpolator's loop (blueice/pdf_morphers.py:57-65): bi_model_begin / _set_anchor / _end
    set_data()  BinnedLogLikelihood.set_data bins the events (likelihood.py:603-609); the histogram goes to bi_upload_counts
    __call__()  names, bounds, priors and the rate pipeline stay here (likelihood.py:328-393,443-481); the interpolation
                of the three tensors, the unphysical-rate test, Beeston-Barlow and the Poisson sum (likelihood.py:355-357,
                397-427,618-675) are ONE bi_eval call

The library is named by the environment variable BLUEICE_HIP_LIB (default: libblueice_hip.so on the loader path).  In the
development container, which has no GPU, tools/run_reference_tests_over_stub.py points it at the host build of the same ABI
(blueice_amd/lib/libblueice_host.so) and runs the reference's own test files over this class.
"""
import ctypes as C
import os

import numpy as np

from blueice.exceptions import NotPreparedException
from blueice.likelihood import BinnedLogLikelihood

BI_ST_OUT_OF_BOUNDS, BI_ST_UNPHYSICAL, BI_ST_BB_ROOT1, BI_ST_BB_NEG = 1, 2, 4, 8
CALLS = dict(bi_eval=0, bi_eval_full=0, bi_model_set_anchor=0, bi_upload_counts=0)     # how often the library was entered

_P = C.c_void_p
_lib = C.CDLL(os.environ.get('BLUEICE_HIP_LIB', 'libblueice_hip.so'))
_lib.bi_create.argtypes = [C.c_int, C.POINTER(_P)]
_lib.bi_destroy.argtypes = [_P]
_lib.bi_destroy.restype = None
_lib.bi_last_error.argtypes = [_P]
_lib.bi_last_error.restype = C.c_char_p
_lib.bi_version.restype = C.c_char_p
_lib.bi_model_begin.argtypes = [_P, C.c_int, _P, _P, C.c_int, C.c_int64, C.c_int]
_lib.bi_model_set_anchor.argtypes = [_P, C.c_int64, _P, _P, _P]
_lib.bi_model_end.argtypes = [_P]
_lib.bi_set_allow_negative.argtypes = [_P, _P]
_lib.bi_upload_counts.argtypes = [_P, C.c_int64, _P]
_lib.bi_eval.argtypes = [_P, C.c_int64, _P, _P, _P, _P, _P]
_lib.bi_eval_full.argtypes = [_P, _P, _P, C.c_int64, _P, _P, _P, _P]
_lib.bi_interpolate.argtypes = [_P, C.c_int, _P, _P]


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def library_version():
    return _lib.bi_version().decode()


class HipBinnedLogLikelihood(BinnedLogLikelihood):

    _h = None

    def _check(self, rc):
        if rc:
            raise RuntimeError(_lib.bi_last_error(self._h).decode())

    def __del__(self):
        if self._h:
            _lib.bi_destroy(self._h)
            self._h = None

    # -- prepare: the anchor tensors --------------------------------------------------------------------------------
    def prepare(self, *args, **kwargs):
        BinnedLogLikelihood.prepare(self, *args, **kwargs)
        if self._h is None:
            self._h = _P()
            if _lib.bi_create(0, C.byref(self._h)):
                raise RuntimeError(_lib.bi_last_error(None).decode())
        self._bb_source = -1
        if self.model_statistical_uncertainty_handling == 'bb_single':
            source_i = self.config.get('bb_single_source')
            if source_i is None:
                raise ValueError("You need to specify bb_single_source to use bb_single_source expectation adjustment")
            self._bb_source = self.base_model.get_source_i(source_i)
        self._S = len(self.source_name_list)
        self._bins = tuple(self.ps.shape[1:])
        self._B = int(np.prod(self._bins))
        if len(self.shape_parameters):
            grids = [np.asarray(g, dtype=float) for g in self.morpher.anchor_z_arrays]
            models = [self.anchor_models[tuple(zs)] for _, zs in self.morpher._anchor_grid_iterator()]
        else:
            grids, models = [], [self.base_model]
        self._d = len(grids)
        n_anchor = np.array([len(g) for g in grids], dtype=np.int32)
        flat = np.concatenate(grids) if grids else np.zeros(0)
        self._check(_lib.bi_model_begin(self._h, self._d, _p(n_anchor), _p(flat), self._S, self._B, self._bb_source))
        for lin, model in enumerate(models):                 # C order over the anchor grid, as pdf_morphers.py:62-65
            ps, n_model = model.pmf_grids()
            ps = np.ascontiguousarray(ps, dtype=float)
            mus = np.ascontiguousarray(model.expected_events(), dtype=float)
            row = np.ascontiguousarray(n_model[self._bb_source], dtype=float) if self._bb_source >= 0 else None
            self._check(_lib.bi_model_set_anchor(self._h, lin, _p(ps), _p(mus), _p(row)))
            CALLS['bi_model_set_anchor'] += 1
        self._check(_lib.bi_model_end(self._h))
        allow = np.array([bool(a) for a in self.source_allowed_negative], dtype=np.int32)
        self._check(_lib.bi_set_allow_negative(self._h, _p(allow)))

    # -- set_data: the binned counts --------------------------------------------------------------------------------
    def set_data(self, d):
        BinnedLogLikelihood.set_data(self, d)                # (prepares trivially when there are no shape parameters)
        n = np.ascontiguousarray(self.data_events_per_bin.histogram, dtype=float)
        self._check(_lib.bi_upload_counts(self._h, 1, _p(n)))
        CALLS['bi_upload_counts'] += 1

    # -- __call__: host half here, device half in the library ----------------------------------------------------
    def __call__(self, livetime_days=None, compute_pdf=False, full_output=False, **kwargs):
        if not self.is_data_set:
            raise NotPreparedException("__call__ requires you to first set the data using set_data()")
        if compute_pdf:                                        # new models on the fly: not the interpolated path
            return BinnedLogLikelihood.__call__(self, livetime_days=livetime_days, compute_pdf=True,
                                                full_output=full_output, **kwargs)
        result = 0
        rate_multipliers, settings = self._kwargs_to_settings(**kwargs)
        zs = []
        for name, (_, log_prior, _) in self.shape_parameters.items():
            z = settings[name]
            zs.append(z)
            minbound, maxbound = self.get_bounds(name)
            if not minbound <= z <= maxbound:
                return -float('inf')
            if log_prior is not None:
                result += log_prior(z)
        z = np.array(zs, dtype=float)
        scale = np.ones(self._S)
        for source_i, source_name in enumerate(self.source_name_list):
            mult = rate_multipliers[source_i]
            scale[source_i] *= mult
            log_prior = self.rate_parameters.get(source_name, None)
            if log_prior is not None:
                result += log_prior(mult)
        if livetime_days is not None:
            if 'livetime_days' not in self.pdf_base_config:
                raise ValueError("Cannot scale live-time, base value absent")
            if self.pdf_base_config['livetime_days'] == 0:
                if livetime_days != 0:
                    raise ValueError("Cannot scale from 0 to non-0 livetime")
                mus = np.empty(self._S)
                self._check(_lib.bi_interpolate(self._h, 1, _p(z), _p(mus)))
                assert np.all(mus * scale == 0), "Got non-0 mus with 0 livetime?!"
            else:
                scale *= livetime_days / self.pdf_base_config['livetime_days']
        if True in self.source_apply_efficiency:
            for i, (sae, sen) in enumerate(zip(self.source_apply_efficiency, self.source_efficiency_names)):
                if sae:
                    scale[i] *= settings.get(sen, 1)
        ll = np.empty(1)
        status = np.zeros(1, dtype=np.int32)
        if full_output:
            mus_out = np.empty(self._S)
            ps_out = np.empty((self._S,) + self._bins)
            self._check(_lib.bi_eval_full(self._h, _p(z), _p(scale), 0, _p(ll), _p(mus_out), _p(ps_out), _p(status)))
            CALLS['bi_eval_full'] += 1
        else:
            self._check(_lib.bi_eval(self._h, 1, _p(z), _p(scale), None, _p(ll), _p(status)))
            CALLS['bi_eval'] += 1
        st = int(status[0])
        if st & BI_ST_OUT_OF_BOUNDS:
            return -float('inf')
        if st & BI_ST_UNPHYSICAL:
            if self.config.get('unphysical_behaviour') == 'error':
                raise ValueError("Unphysical rates (scale %s at %s)" % (scale, z))
            return -float('inf')
        if st & BI_ST_BB_ROOT1:
            raise AssertionError("Beeston-Barlow: the first root is not negative everywhere")      # likelihood.py:649
        if st & BI_ST_BB_NEG:
            raise AssertionError("Beeston-Barlow: negative adjusted expectation")                  # likelihood.py:655
        result += float(ll[0])
        if full_output:
            return result, mus_out, ps_out
        return result
