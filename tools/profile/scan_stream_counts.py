"""Dense-data scans (every bin visited, count-ordered rows: k_scan_sorted) on models whose number of template streams
NS = sources x 2^axes is NOT a power of two: the kernel has one variant per number of 4-stream groups (round 4), so the
matrix work follows ceil(NS / 4), not the next power of two.  python tools/profile/scan_stream_counts.py [out.json]"""
import json, sys, time
sys.path.insert(0, '.')
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
rows = []
ctx = DeviceContext(0)
for S, n_anchor, bins in ((5, (5, 5), (400, 400)), (3, (5, 5), (400, 400)), (6, (5, 5), (400, 400)), (7, (5, 5), (400, 400)),
                          (8, (5, 5), (400, 400)), (3, (4, 4, 4), (60, 60, 60)), (5, (3,), (100000,))):
    m = SyntheticModel(S, n_anchor, bins)
    m.upload(ctx, threads=8)
    NS = S * 2 ** len(n_anchor)
    ctx.set_param('sparse', 0)
    ctx.upload_counts(m.counts(dense=True))
    z, r = m.random_points(65536, seed=11)
    p = ctx.plan(z, r)
    p.run(); ctx.sync()
    ctx.profile(True)
    t = time.perf_counter()
    for _ in range(3): p.run()
    ctx.sync()
    dt = (time.perf_counter() - t) / 3
    n, ms = ctx.profile_read()
    ctx.profile(False)
    groups = (NS + 3) // 4
    rows.append(dict(sources=S, axes=len(n_anchor), streams=NS, stream_groups=groups, bins=m.B, points=len(z), ms_per_scan=dt * 1e3,
                     kernels_ms=ms / 3, evals_per_s=len(z) / dt, TFLOPs_on_padded_groups=2.0 * 4 * groups * m.B * len(z) / dt / 1e12))
    print('S=%d axes=%d NS=%2d (%d groups) B=%d: %.2f ms per scan of %d points, %.0f evaluations/s, %.1f TFLOP/s' % (
        S, len(n_anchor), NS, groups, m.B, dt * 1e3, len(z), len(z) / dt, rows[-1]['TFLOPs_on_padded_groups']), flush=True)
    p.close()
ctx.close()
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(dict(workload='dense-data scans of 65536 points on models with NS not a power of two', command='python tools/profile/scan_stream_counts.py', rows=rows), f, indent=1)
