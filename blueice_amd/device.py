"""Thin object wrapper over the C ABI (include/blueice_hip.h): one `DeviceContext` = one GPU + stream.

Everything numerical happens in libblueice_hip.so; this module only marshals numpy arrays.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import as_f64, ptr
from .exceptions import DeviceError, NotPreparedException

__all__ = ['DeviceContext', 'DeviceBuffer', 'EvalPlan', 'default_device']


def default_device():
    """LOCAL_RANK when launched by torch.distributed.run (one process per GPU), else 0."""
    return int(os.environ.get('BLUEICE_AMD_DEVICE', os.environ.get('LOCAL_RANK', 0)))


class DeviceContext:
    """Owns the anchor tensors and binned data of one likelihood in HBM."""

    def __init__(self, device=None):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        device = default_device() if device is None else int(device)
        rc = self._lib.bi_create(device, C.byref(self._h))
        if rc != 0:
            raise DeviceError("bi_create(device=%d) failed: %s" % (
                device, self._lib.bi_last_error(None).decode()))
        self.device = device
        self._one_out, self._one_st, self._one_ds = C.c_double(), C.c_int32(), C.c_int64()
        self.d = self.S = self.B = None
        self.T = 0
        self.bb_source = -1

    # -- plumbing --------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.bi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == 0:
            return
        msg = self._lib.bi_last_error(self._h).decode()
        if rc == _capi.ERR_STATE:
            raise NotPreparedException(msg)
        if rc == _capi.ERR_INVALID:
            raise ValueError(msg)
        raise DeviceError("libblueice_hip error %d: %s" % (rc, msg))

    def info(self):
        name = C.create_string_buffer(256)
        arch = C.create_string_buffer(256)
        ncu = C.c_int()
        hbm = C.c_int64()
        self._check(self._lib.bi_device_info(self._h, name, arch, 256, C.byref(ncu), C.byref(hbm)))
        return dict(name=name.value.decode(), arch=arch.value.decode(), n_cu=ncu.value, hbm_bytes=hbm.value)

    def set_param(self, name, value):
        self._check(self._lib.bi_set_param(self._h, name.encode(), int(value)))

    def get_param(self, name):
        v = int(self._lib.bi_get_param(self._h, name.encode()))
        if v == -2 ** 63:                      # INT64_MIN: no such (readable) parameter -- never a plausible number
            raise ValueError(self._lib.bi_last_error(self._h).decode())
        return v

    def list_params(self):
        """-> {name: 'rw' | 'r' | 'w'} of every tunable, counter and trigger of the library."""
        n = self._lib.bi_list_params(None, 0)
        buf = C.create_string_buffer(n)
        self._lib.bi_list_params(buf, n)
        return dict(line.split() for line in buf.value.decode().splitlines())

    def sync(self):
        self._check(self._lib.bi_sync(self._h))

    @property
    def stream(self):
        return self._lib.bi_stream(self._h)

    # -- model -----------------------------------------------------------------------------
    def _grid_args(self, anchor_z):
        anchor_z = [np.ascontiguousarray(g, dtype=np.float64) for g in anchor_z]
        n_anchor = np.array([len(g) for g in anchor_z], dtype=np.int32)
        flat = np.concatenate(anchor_z) if len(anchor_z) else np.zeros(0)
        return anchor_z, n_anchor, np.ascontiguousarray(flat, dtype=np.float64)

    def upload_model(self, anchor_z, ps, mus, n_model=None, bb_source=-1):
        """anchor_z: list of d ascending arrays; ps [A.., S, *bins]; mus [A.., S]; n_model like ps."""
        anchor_z, n_anchor, flat = self._grid_args(anchor_z)
        d = len(anchor_z)
        grid_shape = tuple(int(n) for n in n_anchor)
        ps = np.asarray(ps)
        if ps.ndim < d + 2 and not (ps.ndim == d + 1):
            raise ValueError("ps must have shape [A.., S, *bins]")
        S = int(ps.shape[d])
        B = int(np.prod(ps.shape[d + 1:], dtype=np.int64)) if ps.ndim > d + 1 else 1
        ps = as_f64(ps).reshape(grid_shape + (S, B))
        mus = as_f64(mus, grid_shape + (S,))
        if n_model is not None:
            n_model = as_f64(n_model).reshape(grid_shape + (S, B))
        self._check(self._lib.bi_upload_model(self._h, d, ptr(n_anchor), ptr(flat), S, B, ptr(ps), ptr(mus),
                                              ptr(n_model), int(bb_source)))
        self.d, self.S, self.B, self.bb_source = d, S, B, int(bb_source)
        self.anchor_z = anchor_z
        self.T = 0

    def begin_model(self, anchor_z, S, B, bb_source=-1):
        anchor_z, n_anchor, flat = self._grid_args(anchor_z)
        self._check(self._lib.bi_model_begin(self._h, len(anchor_z), ptr(n_anchor), ptr(flat), int(S), int(B),
                                             int(bb_source)))
        self.d, self.S, self.B, self.bb_source = len(anchor_z), int(S), int(B), int(bb_source)
        self.anchor_z = anchor_z
        self.T = 0

    def set_anchor(self, anchor_index, ps, mus, n_model_row=None):
        ps = as_f64(ps).reshape(self.S, self.B)
        mus = as_f64(mus, (self.S,))
        if n_model_row is not None:
            n_model_row = as_f64(n_model_row).reshape(self.B)
        self._check(self._lib.bi_model_set_anchor(self._h, int(anchor_index), ptr(ps), ptr(mus), ptr(n_model_row)))

    def end_model(self):
        self._check(self._lib.bi_model_end(self._h))

    def bb_totals(self, new=None):
        """Per-anchor sums of the Beeston-Barlow source's MC counts; pass `new` to overwrite them with the
        globally reduced values when the bins are sharded over ranks."""
        n_anchor = int(np.prod([len(g) for g in self.anchor_z], dtype=np.int64)) if self.anchor_z else 1
        if new is None:
            out = np.empty(n_anchor, dtype=np.float64)
            self._check(self._lib.bi_get_bb_totals(self._h, ptr(out)))
            return out
        new = as_f64(new, (n_anchor,))
        self._check(self._lib.bi_set_bb_totals(self._h, ptr(new)))
        return new

    def set_allow_negative(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.int32)
        if flags.shape != (self.S,):
            raise ValueError("need one flag per source")
        self._check(self._lib.bi_set_allow_negative(self._h, ptr(flags)))

    # -- data ------------------------------------------------------------------------------
    def upload_counts(self, counts):
        """counts: [*bins] (one dataset) or [T, *bins]."""
        c = as_f64(counts)
        if c.size % self.B:
            raise ValueError("counts size %d is not a multiple of B=%d" % (c.size, self.B))
        T = c.size // self.B
        c = c.reshape(T, self.B)
        self._check(self._lib.bi_upload_counts(self._h, T, ptr(c)))
        self.T = T

    def set_analysis_space(self, edges):
        """edges: one ascending array of bin edges per analysis dimension."""
        edges = [np.ascontiguousarray(e, dtype=np.float64) for e in edges]
        n_edges = np.array([len(e) for e in edges], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(edges))
        self._check(self._lib.bi_set_analysis_space(self._h, len(edges), ptr(n_edges), ptr(flat)))
        self.space_dims = len(edges)

    def upload_events(self, *coords):
        """Bin events (one coordinate array per analysis dimension) on the device into dataset 0."""
        cols = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float64).ravel() for c in coords]))
        if cols.shape[0] != self.space_dims:
            raise ValueError("need %d coordinate arrays" % self.space_dims)
        self._check(self._lib.bi_upload_events(self._h, cols.shape[1], ptr(cols)))
        self.T = 1

    def histogram_events(self, edges, coords):
        """numpy.histogramdd(events, bins=edges)[0] on the device, independent of the context's model and data:
        edges = one ascending array per axis, coords = one coordinate array per axis -> counts [*bins] (float64)."""
        edges = [np.ascontiguousarray(e, dtype=np.float64) for e in edges]
        cols = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float64).ravel() for c in coords]))
        if cols.shape[0] != len(edges):
            raise ValueError("need %d coordinate arrays" % len(edges))
        n_edges = np.array([len(e) for e in edges], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(edges))
        counts = np.empty([len(e) - 1 for e in edges], dtype=np.float64)
        self._check(self._lib.bi_histogram_events(self._h, len(edges), ptr(n_edges), ptr(flat), cols.shape[1], ptr(cols),
                                                  ptr(counts)))
        return counts

    def score_events(self, target, method, grid, coords, outlier_likelihood=1e-12):
        """This context holds density histograms as its model rows: evaluate all of them at the events and make the
        result the (unbinned) model of `target` -- `Model.score_events` for every anchor, on the device.
        method 'piecewise' (grid = bin edges per axis) or 'linear' (grid = bin centres per axis; coords clipped to
        them and finite)."""
        code = {'piecewise': 0, 'linear': 1}[method]
        grid = [np.ascontiguousarray(g, dtype=np.float64) for g in grid]
        cols = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float64).ravel() for c in coords]))
        if cols.shape[0] != len(grid):
            raise ValueError("need %d coordinate arrays" % len(grid))
        n_grid = np.array([len(g) for g in grid], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(grid))
        target._check(self._lib.bi_score_events(self._h, target._h, code, len(grid), ptr(n_grid), ptr(flat), cols.shape[1],
                                                ptr(cols), float(outlier_likelihood)))
        target.d, target.S, target.B, target.bb_source, target.T = self.d, self.S, int(cols.shape[1]), -1, 1
        target.anchor_z = self.anchor_z

    def simulate_events(self, target, method, edges, z, rate_scale=None, seed=0, outlier_likelihood=1e-12):
        """This context holds density histograms as its model rows: draw an event-level toy dataset from them at (z,
        rate_scale) on the device -- `Model.simulate` for histogram-pdf sources -- and make it the (unbinned) data of
        `target`, scored at every anchor model, without a host hop.  edges: the bin edges of every analysis dimension.
        -> events per source [S]."""
        code = {'piecewise': 0, 'linear': 1}[method]
        edges = [np.ascontiguousarray(e, dtype=np.float64) for e in edges]
        n_edges = np.array([len(e) for e in edges], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(edges))
        z = as_f64(z).reshape(self.d) if self.d else None
        if rate_scale is not None:
            rate_scale = as_f64(rate_scale, (self.S,))
        per_source = np.zeros(self.S, dtype=np.int64)
        target._check(self._lib.bi_simulate_events(self._h, target._h, ptr(z), ptr(rate_scale), code, len(edges), ptr(n_edges),
                                                   ptr(flat), int(seed) & (2**64 - 1), float(outlier_likelihood), ptr(per_source)))
        target.d, target.S, target.B, target.bb_source, target.T = self.d, self.S, int(per_source.sum()), -1, 1
        target.anchor_z = self.anchor_z
        target._sim_dims = len(edges)
        return per_source

    def download_events(self):
        """The events of the last simulate_events into this context -> (coords [k, N], source index [N])."""
        n = int(self._lib.bi_simulated_event_count(self._h))
        if n < 0:
            raise NotPreparedException("no simulated events are resident")
        coords = np.empty((self._sim_dims, n), dtype=np.float64)
        source = np.empty(n, dtype=np.int32)
        self._check(self._lib.bi_download_events(self._h, ptr(coords), ptr(source)))
        return coords, source

    def set_unbinned(self, outlier_likelihood=1e-12):
        """Treat the uploaded rows as pdf values at the events: extended unbinned likelihood."""
        self._check(self._lib.bi_set_unbinned(self._h, float(outlier_likelihood)))
        self.T = 1

    def counts_to_dense(self):
        """Device-generated toys (non-empty-bin lists) also as the dense [T][B] array, for the paths that visit every bin
        (Beeston-Barlow, sparse = 0)."""
        self._check(self._lib.bi_counts_to_dense(self._h))

    def generate_toys(self, z, rate_scale=None, T=1, seed=0):
        """Replace the data by T Poisson toy datasets drawn on the device at parameter point (z, rate_scale)."""
        z = as_f64(z).reshape(self.d) if self.d else None
        if rate_scale is not None:
            rate_scale = as_f64(rate_scale, (self.S,))
        self._check(self._lib.bi_generate_toys(self._h, ptr(z), ptr(rate_scale), int(T), int(seed) & (2**64 - 1)))
        self.T = int(T)

    def download_counts(self, t=0):
        out = np.empty(self.B, dtype=np.float64)
        self._check(self._lib.bi_download_counts(self._h, int(t), ptr(out)))
        return out

    # -- evaluation ------------------------------------------------------------------------
    def _point_args(self, z, rate_scale, dataset):
        if self.d:
            z = as_f64(z).reshape(-1, self.d)
            P = len(z)
        else:
            P = 1 if rate_scale is None else len(np.atleast_2d(rate_scale))
            z = None
        if rate_scale is not None:
            rate_scale = np.ascontiguousarray(np.broadcast_to(as_f64(np.atleast_2d(rate_scale)), (P, self.S)))
        if dataset is not None:
            dataset = np.ascontiguousarray(np.broadcast_to(np.asarray(dataset, dtype=np.int64), (P,)))
        return P, z, rate_scale, dataset

    def eval_one(self, z, rate_scale=None, dataset=0):
        """One point, the call `lf(**kw)` makes inside a minimizer: -> (ll, status) as Python scalars, with as
        little marshalling as ctypes allows (z [d] and rate_scale [S] must be float64 arrays or None)."""
        if z is not None and (z.dtype != np.float64 or not z.flags.c_contiguous or z.size != self.d):
            z = as_f64(z).reshape(self.d)
        if rate_scale is not None and (rate_scale.dtype != np.float64 or not rate_scale.flags.c_contiguous
                                       or rate_scale.size != self.S):
            rate_scale = as_f64(rate_scale).reshape(self.S)
        ds = self._one_ds
        ds.value = dataset
        rc = self._lib.bi_eval(self._h, 1, z.ctypes.data if self.d else None,
                               rate_scale.ctypes.data if rate_scale is not None else None, C.addressof(ds),
                               C.addressof(self._one_out), C.addressof(self._one_st))
        if rc:
            self._check(rc)
        return self._one_out.value, self._one_st.value

    def eval(self, z, rate_scale=None, dataset=None):
        """-> (ll [P], status [P]); z [P, d] (or [d]), rate_scale [P, S] or None, dataset [P] or None."""
        P, z, rate_scale, dataset = self._point_args(z, rate_scale, dataset)
        out = np.empty(P, dtype=np.float64)
        status = np.zeros(P, dtype=np.int32)
        self._check(self._lib.bi_eval(self._h, P, ptr(z), ptr(rate_scale), ptr(dataset), ptr(out), ptr(status)))
        return out, status

    def eval_begin(self, z, rate_scale=None, dataset=0):
        """First half of a one-point eval: launch and return.  Collect with eval_end(); in between, evaluate other
        contexts (a sum of likelihoods overlaps its terms this way)."""
        z = as_f64(z).reshape(self.d) if self.d else None
        if rate_scale is not None:
            rate_scale = as_f64(rate_scale, (self.S,))
        self._check(self._lib.bi_eval_begin(self._h, ptr(z), ptr(rate_scale), int(dataset)))

    def eval_end(self):
        """-> (ll, status) of the evaluation started by eval_begin()."""
        out = C.c_double()
        status = C.c_int32()
        self._check(self._lib.bi_eval_end(self._h, C.byref(out), C.byref(status)))
        return out.value, status.value

    def eval_grad(self, z, rate_scale=None, dataset=None):
        """Value and analytic gradient in one pass -> (ll [P], dll/dz [P, d], dll/drate_scale [P, S], status)."""
        P, z, rate_scale, dataset = self._point_args(z, rate_scale, dataset)
        ll = np.empty(P, dtype=np.float64)
        grad = np.empty((P, self.d + self.S), dtype=np.float64)
        status = np.zeros(P, dtype=np.int32)
        self._check(self._lib.bi_eval_grad(self._h, P, ptr(z), ptr(rate_scale), ptr(dataset), ptr(ll), ptr(grad),
                                           ptr(status)))
        return ll, grad[:, :self.d], grad[:, self.d:], status

    def fit_batched(self, P, F, kind, index, z0, scale0, unit, dataset, x0, lo, hi, n_kinks, kinks, gtol, max_iter, x, f, flags, counters):
        """bi_fit_batched: the batched profile-fit engine's loop with this context's likelihood as the objective (all
        arguments are C-contiguous numpy arrays of the right dtype or None; results are written into x, f, flags, counters)."""
        self._check(self._lib.bi_fit_batched(self._h, int(P), int(F), ptr(kind), ptr(index), ptr(z0) if self.d else None, ptr(scale0), ptr(unit),
                                             ptr(dataset), ptr(x0), ptr(lo), ptr(hi), ptr(n_kinks), ptr(kinks), float(gtol), int(max_iter),
                                             ptr(x), ptr(f), ptr(flags), ptr(counters)))
        return 0

    def eval_datasets(self, z, rate_scale=None, t0=0, t1=None):
        """One parameter point against datasets [t0, t1) -> (ll [t1-t0], status)."""
        t1 = self.T if t1 is None else int(t1)
        z = as_f64(z).reshape(self.d) if self.d else None
        if rate_scale is not None:
            rate_scale = as_f64(rate_scale, (self.S,))
        out = np.empty(max(t1 - int(t0), 0), dtype=np.float64)
        status = np.zeros(1, dtype=np.int32)
        self._check(self._lib.bi_eval_datasets(self._h, ptr(z), ptr(rate_scale), int(t0), t1, ptr(out), ptr(status)))
        return out, int(status[0])

    def eval_datasets_device(self, out_ptr, z, rate_scale=None, t0=0, t1=None):
        """eval_datasets with the ll vector left in HBM at device address `out_ptr` -> status."""
        t1 = self.T if t1 is None else int(t1)
        z = as_f64(z).reshape(self.d) if self.d else None
        if rate_scale is not None:
            rate_scale = as_f64(rate_scale, (self.S,))
        status = np.zeros(1, dtype=np.int32)
        self._check(self._lib.bi_eval_datasets_device(self._h, ptr(z), ptr(rate_scale), int(t0), t1, C.c_void_p(out_ptr),
                                                      ptr(status)))
        return int(status[0])

    def eval_datasets_points(self, z, rate_scale=None, t0=0, t1=None, out=None):
        """P parameter points against datasets [t0, t1) in one call (the toy-MC form over several hypotheses: the points of a
        pass share the passes over the templates and over the datasets' lists) -> (ll [P, t1 - t0], status [P]).
        `out`: a C-contiguous float64 array of that shape to write into (a loop that reuses it saves the page faults of a
        fresh 8 P T-byte array per call: 0.1 ms at 32 x 10^4)."""
        t1 = self.T if t1 is None else int(t1)
        P, z, rate_scale, _ = self._point_args(z, rate_scale, None)
        shape = (P, max(t1 - int(t0), 0))
        if out is None:
            out = np.empty(shape, dtype=np.float64)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape %s" % (shape,))
        status = np.zeros(P, dtype=np.int32)
        self._check(self._lib.bi_eval_datasets_points(self._h, P, ptr(z), ptr(rate_scale), int(t0), t1, ptr(out), ptr(status)))
        return out, status

    def eval_datasets_points_device(self, out_ptr, z, rate_scale=None, t0=0, t1=None):
        """eval_datasets_points with the [P, t1 - t0] result left in HBM at device address `out_ptr` -> status [P]."""
        t1 = self.T if t1 is None else int(t1)
        P, z, rate_scale, _ = self._point_args(z, rate_scale, None)
        status = np.zeros(P, dtype=np.int32)
        self._check(self._lib.bi_eval_datasets_points_device(self._h, P, ptr(z), ptr(rate_scale), int(t0), t1, C.c_void_p(out_ptr),
                                                             ptr(status)))
        return status

    def interpolate(self, which, z):
        """which: 'ps' -> [S, B], 'mus' -> [S], 'n_model' -> [B] (the Beeston-Barlow source row)."""
        code = {'ps': 0, 'mus': 1, 'n_model': 2}[which]
        shape = {0: (self.S, self.B), 1: (self.S,), 2: (self.B,)}[code]
        z = as_f64(z).reshape(self.d) if self.d else None
        out = np.empty(shape, dtype=np.float64)
        self._check(self._lib.bi_interpolate(self._h, code, ptr(z), ptr(out)))
        return out

    def eval_full(self, z, rate_scale=None, dataset=0):
        """full_output form -> (ll, mus [S], ps [S, B], status)."""
        z = as_f64(z).reshape(self.d) if self.d else None
        if rate_scale is not None:
            rate_scale = as_f64(rate_scale, (self.S,))
        ll = np.zeros(1)
        mus = np.zeros(self.S)
        ps = np.zeros((self.S, self.B))
        st = np.zeros(1, dtype=np.int32)
        self._check(self._lib.bi_eval_full(self._h, ptr(z), ptr(rate_scale), int(dataset), ptr(ll), ptr(mus), ptr(ps),
                                           ptr(st)))
        return float(ll[0]), mus, ps, int(st[0])

    def plan(self, z, rate_scale=None, dataset=None):
        P, z, rate_scale, dataset = self._point_args(z, rate_scale, dataset)
        h = C.c_void_p()
        self._check(self._lib.bi_plan_points(self._h, P, ptr(z), ptr(rate_scale), ptr(dataset), C.byref(h)))
        return EvalPlan(self, h, P)

    def plan_share(self, z, rate_scale=None, dataset=None, rank=0, world=1):
        """This context's share of a scan dealt over `world` GPUs: every rank passes the same points, the device
        planner's (cell, dataset) sort is the dealing.  -> EvalPlan with `.share = (lo, hi)` and `.n_valid`; its run()
        leaves hi - lo results in sorted order, `unsort()` turns the gathered vectors into the caller's order."""
        P, z, rate_scale, dataset = self._point_args(z, rate_scale, dataset)
        h = C.c_void_p()
        self._check(self._lib.bi_plan_points_share(self._h, P, ptr(z), ptr(rate_scale), ptr(dataset), int(rank), int(world),
                                                   C.byref(h)))
        plan = EvalPlan(self, h, P)
        nv, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.bi_plan_share_info(h, C.byref(nv), C.byref(lo), C.byref(hi)))
        plan.n_valid, plan.share, plan.world = nv.value, (lo.value, hi.value), int(world)
        return plan

    def plan_resident(self, P, z, rate_scale=None, dataset=None, rank=0, world=1):
        """plan() / plan_share() for points that are already in HBM: `z` [P][d] (and `rate_scale` [P][S], `dataset`
        [P] int64) are DeviceBuffers (or device addresses) on this GPU, read where they lie.  world > 1: this rank's
        share of a dealt scan, as plan_share.  Always planned on the device: a batch with infinite rates of a source that may go
        negative, or with Beeston-Barlow points that need the host planner's exact totals, is refused (ValueError)."""
        def addr(b, item_bytes, what):
            if b is None:
                return None
            if isinstance(b, DeviceBuffer):
                if b.nbytes < int(P) * item_bytes:
                    raise ValueError("%s holds %d bytes, %d points need %d" % (what, b.nbytes, P, int(P) * item_bytes))
                return C.c_void_p(b.ptr)
            return C.c_void_p(int(b))
        h = C.c_void_p()
        self._check(self._lib.bi_plan_points_resident(self._h, int(P), addr(z, 8 * max(self.d or 0, 0), 'z'),
                                                      addr(rate_scale, 8 * (self.S or 0), 'rate_scale'), addr(dataset, 8, 'dataset'),
                                                      int(rank), int(world), C.byref(h)))
        plan = EvalPlan(self, h, int(P))
        if world > 1:
            nv, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
            self._check(self._lib.bi_plan_share_info(h, C.byref(nv), C.byref(lo), C.byref(hi)))
            plan.n_valid, plan.share, plan.world = nv.value, (lo.value, hi.value), int(world)
        return plan

    # -- plain device buffers (gather staging) ------------------------------------------------------
    def device_alloc(self, nbytes):
        """-> DeviceBuffer of `nbytes` on this context's GPU.  `.free()` releases it; buffers still alive when the
        context is closed are released with it (the library keeps track of them: `user_allocations`)."""
        h = C.c_void_p()
        self._check(self._lib.bi_device_alloc(self._h, int(nbytes), C.byref(h)))
        return DeviceBuffer(self, h.value, int(nbytes))

    # -- measurement -----------------------------------------------------------------------
    def profile(self, on):
        self._check(self._lib.bi_profile_enable(self._h, 1 if on else 0))

    def read_bandwidth(self, nontemporal=True, blocks_per_cu=32, reps=5):
        """GB/s of a plain streaming sum over the resident template tensor: the device's read-only ceiling."""
        out = C.c_double()
        self._check(self._lib.bi_measure_read_bandwidth(self._h, 1 if nontemporal else 0, int(blocks_per_cu), int(reps),
                                                        C.byref(out)))
        return out.value

    def stream_bandwidth(self, items=8, rows=32, nontemporal=True, blocks_per_cu=8, reps=5):
        """GB/s of `items` work items streaming `rows` template rows each with the morph kernel's access pattern and
        no arithmetic: the ceiling the morph + reduce kernel is held against."""
        out = C.c_double()
        self._check(self._lib.bi_measure_stream_bandwidth(self._h, int(items), int(rows), 1 if nontemporal else 0,
                                                          int(blocks_per_cu), int(reps), C.byref(out)))
        return out.value

    def copy_bandwidth(self, nbytes=1 << 31, reps=3):
        """GB/s (read + written) of a device-to-device copy of the first `nbytes` of the template tensor."""
        out = C.c_double()
        self._check(self._lib.bi_measure_copy_bandwidth(self._h, int(nbytes), int(reps), C.byref(out)))
        return out.value

    def profile_read(self):
        n = C.c_int64()
        ms = C.c_double()
        self._check(self._lib.bi_profile_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value


class DeviceBuffer:
    """A plain allocation in HBM: `.ptr` is the device address (an int), e.g. the target of EvalPlan.run or the
    send / receive buffer of an RCCL collective."""

    def __init__(self, ctx, ptr_value, nbytes):
        self.ctx, self.ptr, self.nbytes = ctx, ptr_value, nbytes

    def to_host(self, dtype=np.float64, count=None, offset_bytes=0, out=None):
        """`out`: a C-contiguous array of `count` elements of `dtype` to copy into (a loop that reuses it saves the page faults of a
        fresh array per call: ~0.5 ms per 8 MB, and the runtime's slow path into memory it has not seen)."""
        n = (self.nbytes - offset_bytes) // np.dtype(dtype).itemsize if count is None else int(count)
        if out is None:
            out = np.empty(n, dtype=dtype)
        elif out.size != n or out.dtype != np.dtype(dtype) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous %s array of %d elements" % (np.dtype(dtype), n))
        self.ctx._check(self.ctx._lib.bi_memcpy_to_host(self.ctx._h, ptr(out), C.c_void_p(self.ptr + offset_bytes), out.nbytes))
        return out

    def from_host(self, a, offset_bytes=0):
        a = np.ascontiguousarray(a)
        if a.nbytes + offset_bytes > self.nbytes:
            raise ValueError("array does not fit the device buffer")
        self.ctx._check(self.ctx._lib.bi_memcpy_to_device(self.ctx._h, C.c_void_p(self.ptr + offset_bytes), ptr(a), a.nbytes))

    def free(self):
        if self.ptr and self.ctx._h.value:
            self.ctx._lib.bi_device_free(self.ctx._h, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class EvalPlan:
    """A batch of points whose host-side preparation (cell lookup, rates, grouping) is done and
    resident on the device; `run()` only launches kernels."""

    def __init__(self, ctx, handle, P):
        self.ctx, self._h, self.P = ctx, handle, P

    @property
    def bytes(self):
        return int(self.ctx._lib.bi_plan_bytes(self._h))

    @property
    def launches(self):
        return int(self.ctx._lib.bi_plan_launches(self._h))

    def run(self, out_dev_ptr=None):
        self.ctx._check(self.ctx._lib.bi_run_plan(self.ctx._h, self._h, out_dev_ptr))
        return self

    def unsort(self, gathered_ptr, stride, full_ptr):
        """A share of a dealt scan: gathered [world][stride] (device; rank r's results in sorted order) -> full [P] in
        the caller's point order (device), on the context stream."""
        self.ctx._check(self.ctx._lib.bi_plan_unsort(self.ctx._h, self._h, C.c_void_p(gathered_ptr), int(stride),
                                                     C.c_void_p(full_ptr)))

    def status(self):
        """Bitwise OR of the per-point status words of the last run() (waits for it) -- for callers that leave the
        results in HBM.  Raises DeviceError on BI_ST_INTERNAL (a launch gave up collecting a partial sum: the
        results of that run are not to be used; the library has emptied its mailbox again)."""
        word = C.c_int32()
        self.ctx._check(self.ctx._lib.bi_plan_status(self.ctx._h, self._h, C.byref(word)))
        if word.value & _capi.ST_INTERNAL:
            raise DeviceError("a launch of this plan gave up waiting for a partial sum (BI_ST_INTERNAL): results discarded")
        return word.value

    def read(self):
        out = np.empty(self.P, dtype=np.float64)
        status = np.zeros(self.P, dtype=np.int32)
        self.ctx._check(self.ctx._lib.bi_plan_read(self.ctx._h, self._h, ptr(out), ptr(status)))
        return out, status

    def close(self):
        if self._h is not None and self._h.value and self.ctx._h.value:
            self.ctx._lib.bi_plan_destroy(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
