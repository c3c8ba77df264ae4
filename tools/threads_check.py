"""Two Python threads, one context each, evaluating concurrently (ctypes releases the interpreter lock during the
calls): results must equal the single-threaded ones."""
import sys, threading
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
models = [SyntheticModel.named('mini3', seed=s) for s in (1, 2, 3, 4)]
ctxs, pts, want = [], [], []
for m in models:
    c = DeviceContext(0); m.upload(c); c.upload_counts(m.counts(dense=True))
    z, r = m.random_points(900, seed=9)
    ctxs.append(c); pts.append((z, r)); want.append((c.eval(z, r)[0], np.array([c.eval(z[i], r[i])[0][0] for i in range(50)])))
errors = []
def work(k):
    c, (z, r) = ctxs[k], pts[k]
    for it in range(200):
        got = c.eval(z, r)[0]
        if not np.array_equal(got, want[k][0]): errors.append((k, it, 'batch'))
        one = np.array([c.eval(z[i], r[i])[0][0] for i in range(50)])
        if not np.array_equal(one, want[k][1]): errors.append((k, it, 'single'))
        g = c.eval_grad(z[:3], r[:3])[0]
        if not np.allclose(g, want[k][0][:3], rtol=1e-12): errors.append((k, it, 'grad'))
threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
for t in threads: t.start()
for t in threads: t.join()
print('4 threads x 200 rounds (batch of 900, 50 single calls, gradient): %d mismatches' % len(errors), errors[:5])
assert not errors
