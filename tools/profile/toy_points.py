"""GPU box: the toy-MC call over several hypotheses (bi_eval_datasets_points: 10^4 device-drawn datasets of C2, P parameter
points per call, results on the host) against P calls of bi_eval_datasets: wall time per call and kernel time by HIP events,
for points in ONE grid cell (rate hypotheses: one shared pass over the templates) and in different cells, by points per pass
and lanes per run.
python tools/profile/toy_points.py [calls] [out.json]"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
T = 10000
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
z, r = m.default_point()
ctx.set_param('sparse', 1)
ctx.generate_toys(z, r, T, seed=4242)
rows = []


def points(P, same_cell, k=0):
    if same_cell:
        zs = np.repeat((z + 0.001 * (k % 50))[None, :], P, axis=0)
        rs = np.repeat(r[None, :], P, axis=0)
        rs[:, 0] *= np.linspace(0.5, 2.0, P)
    else:
        zs, rs = m.random_points(P, seed=100 + k % 7)
    return np.ascontiguousarray(zs), np.ascontiguousarray(rs)


def timed(fn, reps):
    for k in range(3):
        fn(k)
    t = time.perf_counter()
    for k in range(reps):
        fn(k)
    dt = (time.perf_counter() - t) / reps
    ctx.profile(True)
    for k in range(10):
        fn(k)
    launches, ms = ctx.profile_read(); ctx.profile(False)
    return dt * 1e3, ms / 10


one_ms, one_k = timed(lambda k: ctx.eval_datasets(z + 0.001 * (k % 50), r), n)
print('bi_eval_datasets (one point): %.4f ms per call wall = %.1f M evaluations/s, kernels %.4f ms' % (one_ms, T / one_ms / 1e3, one_k), flush=True)
rows.append(dict(call='bi_eval_datasets', P=1, ms_per_call=one_ms, kernels_ms=one_k, evals_per_s=T / one_ms * 1e3))
for same in (True, False):
    for pp, lanes in ((4, 0), (4, 2), (4, 8), (2, 0)):
        ctx.set_param('toy_points_pp', pp)
        ctx.set_param('toy_points_lanes', lanes)
        for P in ((2, 4, 8, 32) if (pp, lanes) == (4, 0) else (4, 32)):
            ms, kms = timed(lambda k: ctx.eval_datasets_points(*points(P, same, k)), max(5, n // P))
            # (the argument arrays are made inside the call: ~10 us of numpy per call, part of what a caller pays too)
            print('%s cell%s, %2d points per call, %d per pass, lanes %d: %.4f ms per call wall = %.1f M evaluations/s (x%.2f per GPU vs one point per call), kernels %.4f ms' % (
                'one' if same else 'random', ' ' if same else 's', P, pp, lanes, ms, P * T / ms / 1e3, (P * T / ms) / (T / one_ms), kms), flush=True)
            rows.append(dict(call='bi_eval_datasets_points', same_cell=same, P=P, points_per_pass=pp, lanes=lanes, ms_per_call=ms,
                             kernels_ms=kms, evals_per_s=P * T / ms * 1e3, speedup_vs_single_point_calls=(P * T / ms) / (T / one_ms)))
ctx.set_param('toy_points_pp', 0)
ctx.set_param('toy_points_lanes', 0)
ctx.close()
if len(sys.argv) > 2:
    with open(sys.argv[2], 'w') as f:
        json.dump(dict(workload='10^4 toy datasets of C2 (configs[2]), P parameter points per call, results to the host',
                       command='python tools/profile/toy_points.py', rows=rows), f, indent=1)
