"""GPU box: the synchronous toy-MC call of configs[2] (bi_eval_datasets: 10^4 device-drawn datasets of C2, one parameter
point per call, results on the host) with the pieces of round 4's short call switched on one by one:
toy_fast_call bits 1 = descriptors in the kernel arguments, 2 = parallel finish, 4 = poll the completion word.
python tools/profile/toy_call_variants.py [calls] [out.json]"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
from blueice_amd.device import DeviceContext
from blueice_amd.synthetic import SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = SyntheticModel.named('C2')
ctx = DeviceContext(0)
m.upload(ctx, threads=8)
z, r = m.default_point()
ctx.set_param('sparse', 1)
ctx.generate_toys(z, r, 10000, seed=4242)
rows = []
ref = None
for bpc in (1, 2, 4, 0):
    ctx.set_param('dot_blocks_per_cu', bpc)
    for k in range(5):
        ctx.eval_datasets(z + 0.01 * k, r)
    t = time.perf_counter()
    for k in range(n):
        ctx.eval_datasets(z + 0.001 * (k % 50), r)
    dt = (time.perf_counter() - t) / n
    ctx.profile(True)
    for k in range(20):
        ctx.eval_datasets(z + 0.001 * (k % 50), r)
    launches, ms = ctx.profile_read(); ctx.profile(False)
    print('dot_blocks_per_cu %d: %.4f ms per call wall, kernels %.4f ms by HIP events' % (bpc, dt * 1e3, ms / 20), flush=True)
for bits in (0, 1, 3, 7, 0, 7):
    ctx.set_param('toy_fast_call', bits)
    for k in range(5):
        ctx.eval_datasets(z + 0.01 * k, r)
    t = time.perf_counter()
    for k in range(n):
        res, st = ctx.eval_datasets(z + 0.001 * (k % 50), r)
    dt = (time.perf_counter() - t) / n
    ctx.profile(True)
    for k in range(20):
        ctx.eval_datasets(z + 0.001 * (k % 50), r)
    launches, ms = ctx.profile_read(); ctx.profile(False)
    chk, _ = ctx.eval_datasets(z, r)
    if ref is None:
        ref = chk
    rows.append(dict(toy_fast_call=bits, ms_per_call=dt * 1e3, kernels_ms_per_call=ms / 20, evals_per_s=1e4 / dt,
                     max_rel_diff_to_bits_0=float(np.max(np.abs(chk - ref) / np.abs(ref)))))
    print('toy_fast_call %d: %.4f ms per call wall (%.1f M evaluations/s), kernels %.4f ms by HIP events; max rel diff to bits 0 %.1e' % (
        bits, dt * 1e3, 1e4 / dt / 1e6, ms / 20, rows[-1]['max_rel_diff_to_bits_0']), flush=True)
ctx.close()
if len(sys.argv) > 2:
    with open(sys.argv[2], 'w') as f:
        json.dump(dict(workload='bi_eval_datasets: 10^4 toy datasets of C2, one point per call, results to the host', command='python tools/profile/toy_call_variants.py', rows=rows), f, indent=1)
