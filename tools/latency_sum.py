"""A LogLikelihoodSum of three device likelihoods: overlapped (begin on all, end on all) vs one after the other."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd import LogLikelihoodSum
from blueice_amd.synthetic import SyntheticModel
name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
terms = []
for s in (1, 2, 3):
    m = SyntheticModel.named(name, seed=s)
    lf = m.likelihood()
    lf.set_binned_data(m.counts().reshape(m.bins))
    terms.append(lf)
tot = LogLikelihoodSum(terms)
kw = dict(shape0=0.3, s1_rate_multiplier=1.1)
assert tot(**kw) == sum(lf(**kw) for lf in terms)
for label, f in (('overlapped sum', lambda: tot(**kw)), ('terms one by one', lambda: sum(lf(**kw) for lf in terms))):
    f()
    ts = []
    for _ in range(1000):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    print('%s, %-17s median %.1f us' % (name, label, np.median(ts) * 1e6), flush=True)
