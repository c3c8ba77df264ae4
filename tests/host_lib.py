"""ctypes binding of libblueice_host.so -- the host build of the C ABI's minimal entry points (blueice_amd/csrc/
host_backend.cpp).  Test infrastructure: the package itself never loads that library."""
import ctypes as C

import numpy as np

_P = C.c_void_p


def load(path=None):
    """path = None: the host build; tests/test_minimal_abi_gpu.py passes libblueice_hip.so -- the SAME signatures, the product."""
    from blueice_amd import build
    lib = C.CDLL(path or build.build_host())
    lib.bi_create.argtypes = [C.c_int, C.POINTER(_P)]
    lib.bi_destroy.argtypes = [_P]
    lib.bi_destroy.restype = None
    lib.bi_last_error.argtypes = [_P]
    lib.bi_last_error.restype = C.c_char_p
    lib.bi_version.restype = C.c_char_p
    lib.bi_upload_model.argtypes = [_P, C.c_int, _P, _P, C.c_int, C.c_int64, _P, _P, _P, C.c_int]
    lib.bi_model_begin.argtypes = [_P, C.c_int, _P, _P, C.c_int, C.c_int64, C.c_int]
    lib.bi_model_set_anchor.argtypes = [_P, C.c_int64, _P, _P, _P]
    lib.bi_model_end.argtypes = [_P]
    lib.bi_set_allow_negative.argtypes = [_P, _P]
    lib.bi_upload_counts.argtypes = [_P, C.c_int64, _P]
    lib.bi_eval.argtypes = [_P, C.c_int64, _P, _P, _P, _P, _P]
    lib.bi_eval_full.argtypes = [_P, _P, _P, C.c_int64, _P, _P, _P, _P]
    lib.bi_interpolate.argtypes = [_P, C.c_int, _P, _P]
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


class HostContext:
    """The few calls the tests need, with numpy arrays."""

    def __init__(self, lib):
        self.lib = lib
        self.h = _P()
        assert lib.bi_create(0, C.byref(self.h)) == 0

    def close(self):
        if self.h:
            self.lib.bi_destroy(self.h)
            self.h = _P()

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.lib.bi_last_error(self.h).decode())

    def upload_model(self, anchor_z, ps, mus, n_model=None, bb_source=-1, allow_negative=None):
        d = len(anchor_z)
        self.d = d
        n_anchor = np.array([len(g) for g in anchor_z], np.int32)
        flat = np.concatenate([np.asarray(g, float) for g in anchor_z]) if d else np.zeros(0)
        ps = np.ascontiguousarray(ps, float)
        mus = np.ascontiguousarray(mus, float)
        self.S = mus.shape[-1]
        self.bins = ps.shape[d + 1:]
        self.B = int(np.prod(self.bins))
        nm = None if n_model is None or bb_source < 0 else np.ascontiguousarray(n_model, float)
        self._check(self.lib.bi_upload_model(self.h, d, ptr(n_anchor), ptr(flat), self.S, self.B, ptr(ps), ptr(mus), ptr(nm),
                                             bb_source if nm is not None else -1))
        if allow_negative is not None:
            self._check(self.lib.bi_set_allow_negative(self.h, ptr(np.array(allow_negative, np.int32))))

    def upload_counts(self, counts):
        c = np.ascontiguousarray(counts, float).reshape(-1, self.B)
        self._check(self.lib.bi_upload_counts(self.h, c.shape[0], ptr(c)))

    def eval(self, z, scale=None, dataset=None):
        z = np.ascontiguousarray(np.atleast_2d(np.asarray(z, float)).reshape(-1, max(self.d, 1))[:, :self.d] if self.d else
                                 np.zeros((1 if scale is None else np.atleast_2d(scale).shape[0], 0)), float)
        P = z.shape[0] if self.d else (1 if scale is None else np.atleast_2d(scale).shape[0])
        sc = None if scale is None else np.ascontiguousarray(np.atleast_2d(np.asarray(scale, float)), float)
        ds = None if dataset is None else np.ascontiguousarray(dataset, np.int64)
        out = np.empty(P)
        st = np.zeros(P, np.int32)
        self._check(self.lib.bi_eval(self.h, P, ptr(z) if self.d else None, ptr(sc), ptr(ds), ptr(out), ptr(st)))
        return out, st

    def eval_full(self, z, scale=None, dataset=0):
        z = np.ascontiguousarray(z, float)
        sc = None if scale is None else np.ascontiguousarray(scale, float)
        ll = C.c_double()
        st = C.c_int32()
        mus = np.empty(self.S)
        ps = np.empty((self.S,) + tuple(self.bins))
        self._check(self.lib.bi_eval_full(self.h, ptr(z) if self.d else None, ptr(sc), dataset, C.byref(ll), ptr(mus), ptr(ps), C.byref(st)))
        return ll.value, mus, ps, st.value

    def interpolate(self, which, z):
        shape = {0: (self.S,) + tuple(self.bins), 1: (self.S,), 2: tuple(self.bins)}[which]
        out = np.empty(shape)
        self._check(self.lib.bi_interpolate(self.h, which, ptr(np.ascontiguousarray(z, float)) if self.d else None, ptr(out)))
        return out
