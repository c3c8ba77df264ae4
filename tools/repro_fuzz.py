"""Re-run one seed of tests/test_fuzz_gpu.py::test_large_batches_... and print every mismatch in full."""
import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
seed = int(sys.argv[1])
os.environ['BLUEICE_FUZZ_SEEDS'] = '%d:1' % seed
import test_fuzz_gpu as t
from blueice_amd.device import DeviceContext
from oracle import blueice_oracle as orc
RTOL = 1e-10
rng = np.random.default_rng(5000 + seed)
ctx = DeviceContext(0)
for rep in range(3):
    d = int(rng.integers(0, 4)); S = int(rng.choice([1, 2, 3, 4, 5, 8])); B = int(rng.choice([40, 511, 512, 700, 1300]))
    model, counts0 = t.random_case(rng, d, S, B, -1)
    T = int(rng.integers(1, 4))
    counts = np.stack([rng.poisson(counts0 * rng.uniform(0.5, 2)).astype(float) for _ in range(T)])
    if rng.random() < 0.5:
        counts[rng.integers(T), rng.integers(B)] = rng.choice([np.nan, -1.0, 2.5])
    allow_negative = None
    if rng.random() < 0.4:
        allow_negative = np.zeros(S, dtype=bool); allow_negative[rng.integers(S)] = True
    P = int(rng.integers(600, 1500))
    z, r = t.random_points(rng, model, P, S)
    if allow_negative is not None:
        neg = rng.random(P) < 0.2
        r[neg, np.flatnonzero(allow_negative)[0]] = -rng.uniform(0.5, 3.0, neg.sum())
    ds = rng.integers(0, T, P)
    bad = rng.random(P)
    if d:
        z[bad < 0.02, 0] = 99.0
        z[(bad > 0.02) & (bad < 0.03), d - 1] = np.nan
    r[(bad > 0.03) & (bad < 0.05), 0] = -0.5 if allow_negative is None or not allow_negative[0] else np.inf
    ds[(bad > 0.05) & (bad < 0.06)] = T + 3
    want = np.array([orc.loglikelihood(model, counts[ds[i]], z[i], r[i], allow_negative=allow_negative) if 0 <= ds[i] < T else np.nan for i in range(P)])
    ctx.upload_model(model['anchor_z'], model['ps'], model['mus'])
    if allow_negative is not None: ctx.set_allow_negative(allow_negative)
    for sparse in (0, 1):
        ctx.set_param('sparse', sparse); ctx.upload_counts(counts)
        before = ctx.get_param('n_scan_launches')
        got, st = ctx.eval(z if d else None, r, dataset=ds)
        used = ctx.get_param('n_scan_launches') > before
        nbad = 0
        for i in range(P):
            if not 0 <= ds[i] < T: continue
            w, g = want[i], got[i]
            ok = (np.isnan(w) and np.isnan(g)) or g == w or (np.isfinite(w) and abs(g - w) <= RTOL * max(1, abs(w)))
            if not ok:
                nbad += 1
                if nbad <= 5:
                    print('rep %d d=%d S=%d B=%d T=%d sparse=%d scan=%d allow_neg=%s point %d: got %r want %r status %d z=%s r=%s ds=%d' % (
                        rep, d, S, B, T, sparse, used, allow_negative, i, g, w, st[i], z[i], r[i], ds[i]))
                    ctx.set_param('scan_mfma', 0)
                    g2, _ = ctx.eval(z if d else None, r, dataset=ds)
                    ctx.set_param('scan_mfma', 1)
                    one, st1 = ctx.eval(z[i] if d else None, r[i], dataset=np.array([ds[i]]))
                    print('     vector kernel: %r   single call: %r (status %d)' % (g2[i], one[0], st1[0]))
        print('rep %d sparse=%d: %d mismatches of %d' % (rep, sparse, nbad, P), flush=True)
    if allow_negative is not None: ctx.set_allow_negative(np.zeros(S, dtype=bool))
