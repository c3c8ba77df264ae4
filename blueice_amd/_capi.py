"""ctypes binding of include/blueice_hip.h (libblueice_hip.so).

There is deliberately no fallback: if the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"`, or `python -m blueice_amd.build`) or no
HIP device is present, the first use raises DeviceError.
"""
import ctypes as C
import os

import numpy as np

from .exceptions import DeviceError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BLUEICE_AMD_LIB') or os.path.join(_HERE, 'lib', 'libblueice_hip.so')   # env: A/B builds

# status bits (include/blueice_hip.h)
ST_OUT_OF_BOUNDS, ST_UNPHYSICAL, ST_BB_ROOT1, ST_BB_NEG, ST_BAD_DATASET, ST_INTERNAL = 1, 2, 4, 8, 16, 32
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_NOMEM = -1, -2, -3, -4

_p = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double
_pd = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    'bi_create': (C.c_int, [C.c_int, C.POINTER(_p)]),
    'bi_destroy': (None, [_p]),
    'bi_last_error': (C.c_char_p, [_p]),
    'bi_version': (C.c_char_p, []),
    'bi_device_info': (C.c_int, [_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_i64)]),
    'bi_upload_model': (C.c_int, [_p, C.c_int, _p, _p, C.c_int, _i64, _p, _p, _p, C.c_int]),
    'bi_model_begin': (C.c_int, [_p, C.c_int, _p, _p, C.c_int, _i64, C.c_int]),
    'bi_model_set_anchor': (C.c_int, [_p, _i64, _p, _p, _p]),
    'bi_model_end': (C.c_int, [_p]),
    'bi_get_bb_totals': (C.c_int, [_p, _p]),
    'bi_set_bb_totals': (C.c_int, [_p, _p]),
    'bi_set_allow_negative': (C.c_int, [_p, _p]),
    'bi_upload_counts': (C.c_int, [_p, _i64, _p]),
    'bi_set_analysis_space': (C.c_int, [_p, C.c_int, _p, _p]),
    'bi_upload_events': (C.c_int, [_p, _i64, _p]),
    'bi_histogram_events': (C.c_int, [_p, C.c_int, _p, _p, _i64, _p, _p]),
    'bi_score_events': (C.c_int, [_p, _p, C.c_int, C.c_int, _p, _p, _i64, _p, C.c_double]),
    'bi_set_unbinned': (C.c_int, [_p, _f64]),
    'bi_simulate_events': (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, _p, _p, C.c_uint64, C.c_double, _p]),
    'bi_download_events': (C.c_int, [_p, _p, _p]),
    'bi_simulated_event_count': (_i64, [_p]),
    'bi_generate_toys': (C.c_int, [_p, _p, _p, _i64, C.c_uint64]),
    'bi_download_counts': (C.c_int, [_p, _i64, _p]),
    'bi_eval': (C.c_int, [_p, _i64, _p, _p, _p, _p, _p]),
    'bi_eval_grad': (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p]),
    'bi_minimize_batched': (C.c_int, [_p, _p, _i64, C.c_int, _p, _p, _p, _p, _p, C.c_double, C.c_int, _p, _p, _p, _p]),
    'bi_fit_batched': (C.c_int, [_p, _i64, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.c_double, C.c_int, _p, _p, _p, _p]),
    'bi_eval_datasets': (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    'bi_eval_datasets_points': (C.c_int, [_p, _i64, _p, _p, _i64, _i64, _p, _p]),
    'bi_eval_datasets_points_device': (C.c_int, [_p, _i64, _p, _p, _i64, _i64, _p, _p]),
    'bi_interpolate': (C.c_int, [_p, C.c_int, _p, _p]),
    'bi_eval_full': (C.c_int, [_p, _p, _p, _i64, _p, _p, _p, _p]),
    'bi_plan_points': (C.c_int, [_p, _i64, _p, _p, _p, C.POINTER(_p)]),
    'bi_run_plan': (C.c_int, [_p, _p, _p]),
    'bi_plan_points_share': (C.c_int, [_p, _i64, _p, _p, _p, C.c_int, C.c_int, C.POINTER(_p)]),
    'bi_plan_points_resident': (C.c_int, [_p, _i64, _p, _p, _p, C.c_int, C.c_int, C.POINTER(_p)]),
    'bi_plan_share_info': (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    'bi_plan_unsort': (C.c_int, [_p, _p, _p, _i64, _p]),
    'bi_plan_read': (C.c_int, [_p, _p, _p, _p]),
    'bi_plan_status': (C.c_int, [_p, _p, C.POINTER(_i32)]),
    'bi_plan_bytes': (_i64, [_p]),
    'bi_plan_launches': (_i64, [_p]),
    'bi_plan_destroy': (None, [_p, _p]),
    'bi_sync': (C.c_int, [_p]),
    'bi_stream': (_p, [_p]),
    'bi_eval_begin': (C.c_int, [_p, _p, _p, _i64]),
    'bi_eval_end': (C.c_int, [_p, _p, _p]),
    'bi_eval_datasets_device': (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    'bi_counts_to_dense': (C.c_int, [_p]),
    'bi_device_alloc': (C.c_int, [_p, _i64, C.POINTER(_p)]),
    'bi_device_free': (C.c_int, [_p, _p]),
    'bi_memcpy_to_host': (C.c_int, [_p, _p, _p, _i64]),
    'bi_memcpy_to_device': (C.c_int, [_p, _p, _p, _i64]),
    'bi_selftest_log': (C.c_int, [_p, _i64, _p, _p]),
    'bi_selftest_sort': (C.c_int, [_p, C.c_int, _i64, _p, _p, C.c_int, C.c_int, _p, _p]),
    'bi_selftest_scan': (C.c_int, [_p, C.c_int, _i64, _p, _i64, _p]),
    'bi_measure_read_bandwidth': (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _p]),
    'bi_measure_stream_bandwidth': (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    'bi_measure_copy_bandwidth': (C.c_int, [_p, _i64, C.c_int, _p]),
    'bi_profile_enable': (C.c_int, [_p, C.c_int]),
    'bi_profile_read': (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_f64)]),
    'bi_set_param': (C.c_int, [_p, C.c_char_p, _i64]),
    'bi_get_param': (_i64, [_p, C.c_char_p]),
    'bi_list_params': (C.c_int, [C.c_char_p, C.c_int]),
}

# the objective callback of bi_minimize_batched
OBJECTIVE_FN = C.CFUNCTYPE(C.c_int, _p, _i64, C.c_int, _pd, C.POINTER(_i64), _pd, _pd)

_lib = None


def load():
    """Load (once) and return the shared library with all prototypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeviceError("%s is missing: build it first (python -m blueice_amd.build). "
                          "blueice_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise DeviceError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise DeviceError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    # BLUEICE_AMD_LIB exists for A/B runs of two builds of THIS library; it is not a backend switch: whatever is loaded must
    # say it is the gfx950 build (the host build of the boundary tests calls itself "... not the product")
    version = (lib.bi_version() or b'').decode()
    if not version.startswith('blueice_hip') or '(gfx950)' not in version or 'not the product' in version:
        raise DeviceError("%s is not a gfx950 build of libblueice_hip (bi_version: %r)" % (LIB_PATH, version))
    _lib = lib
    return lib


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected array of shape %s, got %s" % (tuple(shape), a.shape))
    return a
