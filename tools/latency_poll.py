import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
ctx = DeviceContext(0)
m = SyntheticModel.named('C2')
m.upload(ctx)
ctx.upload_counts(m.counts())
z1, r1 = m.random_points(1, seed=3)
ref = {}
for rep in range(4):
    for poll in (0, 1):
        ctx.set_param('poll_result', poll)
        for sp in (1,):
            ctx.set_param('sparse', sp)
            v = ctx.eval(z1, r1)[0][0]
            ref.setdefault(sp, v)
            assert v == ref[sp]
            ts = []
            for _ in range(4000):
                t = time.perf_counter(); ctx.eval(z1, r1); ts.append(time.perf_counter() - t)
            ts = np.array(ts) * 1e6
            print('poll_result=%d sparse=%d: mean %.1f us, median %.1f, p99 %.1f, max %.0f' % (poll, sp, ts.mean(), np.median(ts), np.percentile(ts, 99), ts.max()), flush=True)
