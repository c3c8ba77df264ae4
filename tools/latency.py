"""Per-call latency of the synchronous single-point path (what bestfit_scipy sees)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
import ctypes as C
from blueice_amd import _capi

ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
counts = m.counts()
z, r = m.random_points(64, seed=1)
for sparse in (0, 1):
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(counts)
    for i in range(20): ctx.eval(z[i], r[i])
    t = time.perf_counter()
    for i in range(500): ctx.eval(z[i % 64], r[i % 64])
    dt = (time.perf_counter() - t) / 500
    # raw C call without numpy marshalling
    lib = ctx._lib
    out = np.zeros(1); st = np.zeros(1, np.int32)
    zz = np.ascontiguousarray(z); rr = np.ascontiguousarray(r)
    t = time.perf_counter()
    for i in range(500):
        lib.bi_eval(ctx._h, 1, zz[i % 64].ctypes.data_as(C.c_void_p), rr[i % 64].ctypes.data_as(C.c_void_p), None,
                    out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p))
    dt2 = (time.perf_counter() - t) / 500
    print('sparse=%d: DeviceContext.eval %.1f us/call, raw bi_eval %.1f us/call' % (sparse, dt * 1e6, dt2 * 1e6))
# small batches and the gradient call (recycled transient buffers)
ctx.set_param('sparse', 0)
ctx.upload_counts(counts)
for P in (4, 16):
    for i in range(5): ctx.eval(z[:P], r[:P])
    t = time.perf_counter()
    for i in range(100): ctx.eval(z[:P], r[:P])
    print('dense batch P=%d: %.1f us/call' % (P, (time.perf_counter() - t) / 100 * 1e6))
for sparse in (0, 1):
    ctx.set_param('sparse', sparse)
    ctx.upload_counts(counts)
    for i in range(5): ctx.eval_grad(z[0], r[0])
    t = time.perf_counter()
    for i in range(100): ctx.eval_grad(z[i % 64], r[i % 64])
    print('sparse=%d eval_grad: %.1f us/call' % (sparse, (time.perf_counter() - t) / 100 * 1e6))
