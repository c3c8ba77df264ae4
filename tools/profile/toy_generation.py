import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
z, r = m.default_point()
ctx.set_param('sparse', 1)
for ev in (1, 0):
    ctx.set_param('toy_events', ev)
    ctx.generate_toys(z, r, 10000, seed=4242)
    t = time.perf_counter()
    for k in range(3):
        ctx.generate_toys(z, r, 10000, seed=4242 + k)
    dt = (time.perf_counter() - t) / 3
    ll, st = ctx.eval_datasets(z, r)
    print('10^4 toys, %s: %.1f ms per ensemble (method %d, %d non-empty bins in all); mean ll %.4f +- %.4f' % (
        'event by event' if ev else 'bin by bin', dt * 1e3, ctx.get_param('last_toy_method'), ctx.get_param('nnz_total'), ll.mean(), ll.std() / 100))
