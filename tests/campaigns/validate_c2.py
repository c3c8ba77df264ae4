"""Full-size parity campaign: N random C2 points (every path: dense vector kernel, matrix-core scan kernel,
non-empty-bin form, single calls) against the CPU oracle evaluated on the same synthetic tensors.
The oracle values are computed first, in forked worker processes, before this process touches the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from blueice_amd.synthetic import SyntheticModel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = SyntheticModel.named('C2')
z, r = m.random_points(N, seed=99)
z[: N // 8] = np.round(z[: N // 8])                  # some points exactly on anchors
z[N // 8: N // 6, 0] = 2.0                           # top edge of an axis
r[::17, 1] = 0.0                                     # a switched-off source
DATA = {'sparse': m.counts(), 'dense': m.counts(dense=True)}


def oracle_chunk(args):
    from oracle import blueice_oracle as orc
    lo, hi = args
    out = np.empty((hi - lo, 2))
    for i in range(lo, hi):
        cell = m.cell_model(z[i])
        out[i - lo, 0] = orc.loglikelihood(cell, DATA['sparse'], z[i], r[i])
        out[i - lo, 1] = orc.loglikelihood(cell, DATA['dense'], z[i], r[i])
    return out


if __name__ == '__main__':
    import multiprocessing as mp
    workers = min(16, os.cpu_count() or 1)
    t = time.perf_counter()
    edges = np.linspace(0, N, 4 * workers + 1).astype(int)
    with mp.get_context('fork').Pool(workers) as pool:
        want = np.concatenate(pool.map(oracle_chunk, list(zip(edges[:-1], edges[1:]))))
    print('oracle: %d points x 2 datasets in %.0f s on %d processes' % (N, time.perf_counter() - t, workers), flush=True)

    from blueice_amd.device import DeviceContext
    ctx = DeviceContext(0)
    m.upload(ctx, threads=8)
    worst = 0.0
    for col, (kind, counts) in enumerate(DATA.items()):
        for sparse in (0, 1):
            ctx.set_param('sparse', sparse)
            ctx.upload_counts(counts)
            for label, setup in (('vector kernel', dict(scan_mfma=0, scan_split=0)),
                                 ('matrix-core scan, per-bin terms', dict(scan_mfma=1, scan_min_items=1, scan_split=0)),
                                 ('matrix-core scan, split', dict(scan_mfma=1, scan_min_items=1, scan_split=1))):
                for k, v in setup.items():
                    ctx.set_param(k, v)
                before = (ctx.get_param('n_scan_launches'), ctx.get_param('n_valid_launches'))
                got, st = ctx.eval(z, r)
                scan = ctx.get_param('n_scan_launches') - before[0]
                valid = ctx.get_param('n_valid_launches') - before[1]
                err = np.max(np.abs(got - want[:, col]) / np.abs(want[:, col]))
                worst = max(worst, err)
                print('%-6s data, sparse=%d, %-32s (k_scan_mfma launches %d, k_scan_valid launches %d): max rel diff %.2e, status bits %d' % (
                    kind, sparse, label, scan, valid, err, int(np.bitwise_or.reduce(st))), flush=True)
            single = np.array([ctx.eval(z[i], r[i])[0][0] for i in range(0, N, 8)])
            err = np.max(np.abs(single - want[::8, col]) / np.abs(want[::8, col]))
            worst = max(worst, err)
            print('%-6s data, sparse=%d, single calls: max rel diff %.2e' % (kind, sparse, err), flush=True)
    ctx.set_param('scan_min_items', 4)
    print('worst relative difference over all paths: %.2e (tolerance 1e-10)' % worst)
    assert worst <= 1e-10
