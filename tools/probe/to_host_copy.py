import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from blueice_amd.device import DeviceContext
ctx = DeviceContext(0)
for mb in (2, 4, 8, 16):
    n = mb * 1024 * 1024 // 8
    buf = ctx.device_alloc(8 * n)
    buf.from_host(np.arange(n, dtype=np.float64))
    host = np.empty(n)
    for label, fn in (('fresh array', lambda: buf.to_host(np.float64, n)), ('reused array', lambda: buf.to_host(np.float64, n, out=host))):
        fn(); fn()
        t = time.perf_counter()
        for _ in range(10):
            fn()
        print('%2d MB, %-12s: %.3f ms per copy' % (mb, label, (time.perf_counter() - t) / 10 * 1e3), flush=True)
    buf.free()
ctx.close()
