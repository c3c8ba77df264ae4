"""Helpers to read the golden fixtures of tests/golden (made by tests/golden/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _all_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz'))
                  if not os.path.basename(p).startswith(('fit_', 'api_', 'profile_')))


def case_names():
    """Binned cases."""
    return [n for n in _all_names() if not n.startswith('unb_')]


def unbinned_case_names():
    return [n for n in _all_names() if n.startswith('unb_')]


def load_case(name):
    f = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    d = int(f['d'])
    model = dict(anchor_z=[f['anchor_z_%d' % i] for i in range(d)],
                 ps=f['ps'], mus=f['mus'],
                 n_model=f['n_model'] if 'n_model' in f.files else None)
    case = dict(name=name, d=d, S=int(f['S']), bins=tuple(int(b) for b in f['bins']),
                model=model, counts=f['counts'], bb_source=int(f['bb_source']),
                livetime_base=float(f['livetime_base']),
                allow_negative=[bool(a) for a in f['allow_negative']] if 'allow_negative' in f.files else None,
                kind=int(f['kind']) if 'kind' in f.files else 0,
                outlier=float(f['outlier']) if 'outlier' in f.files else 0.0,
                call_z=f['call_z'], call_mult=f['call_mult'], call_livetime=f['call_livetime'],
                call_ll=f['call_ll'], raw=f)
    return case


def rate_scale_of(case, j):
    """rate multiplier x livetime scaling of call j (likelihood.py:366-382)."""
    if 'call_scale' in case['raw'].files:
        return np.array(case['raw']['call_scale'][j], dtype=float)
    rs = np.array(case['call_mult'][j], dtype=float)
    lt = case['call_livetime'][j]
    if not np.isnan(lt):
        rs = rs * (lt / case['livetime_base'])
    return rs


def same(a, b, rtol=0.0):
    """Equality that treats nan==nan and +-inf exactly; finite values to rtol*max(1,|b|)."""
    a, b = float(a), float(b)
    if np.isnan(b):
        return np.isnan(a)
    if np.isinf(b):
        return a == b
    return abs(a - b) <= rtol * max(1.0, abs(b))
