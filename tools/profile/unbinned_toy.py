"""Event-level toys of the C2 shape on the device (bi_simulate_events: draw ~10^4 events from the morphed histograms, score them at
every anchor model, make them the unbinned data) and one likelihood evaluation each: time per toy, and the kernels of one toy.
python tools/profile/unbinned_toy.py [toys]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = SyntheticModel.named('C2')
tp = DeviceContext(0)
m.upload(tp, threads=8)                      # the pmf rows stand in for density histograms (uniform bins: the same up to a factor)
edges = [np.linspace(0.0, 1.0, b + 1) for b in m.bins]
z, r = m.default_point()
u = DeviceContext(0)
for k in range(3):
    per = tp.simulate_events(u, 'piecewise', edges, z, r, seed=k)
    u.eval(z, r)
t = time.perf_counter()
for k in range(n):
    per = tp.simulate_events(u, 'piecewise', edges, z, r, seed=100 + k)
t_sim = (time.perf_counter() - t) / n
t = time.perf_counter()
for k in range(n):
    ll, st = u.eval(z, r)
t_eval = (time.perf_counter() - t) / n
print('%d events per toy: simulate + score %.2f ms per toy, one evaluation %.3f ms' % (int(per.sum()), t_sim * 1e3, t_eval * 1e3))
u.close(); tp.close()
