#!/usr/bin/env python3
"""Headline benchmark: binned-likelihood evaluations per second on the BASELINE.json model.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1], "C2"): 4 sources, 3 shape parameters with 5 anchors each (125
anchor models), 100x100x100 analysis bins, one dataset; the anchor tensor (4.0 GB fp64) and the counts
are resident in HBM before the timed region.  One STEP = one batched call evaluating 8 independent
parameter points: for every point the morph+reduce kernel streams the 2^3 * 4 corner templates of its
grid cell plus the counts (264 MB per evaluation) and reduces to a scalar -- 2.1 GB
per step, one kernel launch.  The 8 points of a step lie in grid cells that share no anchor model with
each other (per axis cells {0,2} or {1,3}), so no template byte is used twice within a step, and a step's
2.1 GB is far beyond the 256 MiB Infinity Cache: the number is an HBM-streaming number, and the
algorithmic bytes equal the compulsory traffic.  Successive steps rotate through the 8 parity
combinations.  (A batch that covers ALL 64 cells runs ~1.4x faster per evaluation because neighbouring
cells share corner templates in L2 / Infinity Cache -- reported under extras, not as the headline.)

With N > 1 ranks (launched by torch.distributed.run, one process per GPU) every rank holds a replica of
the tensor and evaluates its own K points (weak scaling, no data-path collective); the per-rank result
vectors are gathered once at the end with RCCL (all_gather), inside the timed region.

The JSON line also carries
  roofline      morph+reduce kernel: algorithmic bytes per launch / HIP-event kernel time vs 8 TB/s
  cpu_baseline  the numpy/scipy oracle (the reference's arithmetic) timed on the host, rank 0, N = 1
  extras        other call shapes of the same path (one point per launch, same-cell repeat, scan batch,
                toy-MC, synchronous call latency)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
POOL = 8


def cpu_baseline_all_cores(config, n_procs, budget_s=10.0):
    """N independent host processes (oracle/cpu_worker.py), each evaluating its own point with the oracle
    (BASELINE.md section 4, step 2).  Plain subprocesses: nothing is forked from this GPU-initialised process."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'cpu_worker.py'), config]
    procs = [subprocess.Popen(cmd + [str(500 + i), str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              env=env, text=True) for i in range(n_procs)]
    rate, done = 0.0, 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=budget_s + 120)
            n, dt = out.split()
            rate += float(n) / float(dt)
            done += 1
        except Exception:
            p.kill()
    return dict(value=rate, unit='evals/s', cores=done, kind='port',
                sample='%d processes x %.0f s of single-thread oracle evaluations, one point each' % (done, budget_s))


def cpu_baseline(model, counts, points, budget_s=20.0):
    """Time the oracle (numpy/scipy restatement of the reference path) on this host, one thread."""
    from oracle import blueice_oracle as orc
    z, r = points
    cm = model.cell_model(z[0])
    c = counts
    orc.loglikelihood(cm, c, z[0], r[0])            # warm
    n, t0 = 0, time.perf_counter()
    while True:
        orc.loglikelihood(cm, c, z[0], r[0])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 200:
            break
    return dict(value=n / dt, unit='evals/s', cores=1, kind='port',
                sample='%d single-thread evaluations of the C2 model at one off-grid point (%.1f s); '
                       'numpy %s oracle = the reference arithmetic' % (n, dt, np.__version__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=40)
    ap.add_argument('--config', default='C2')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo for\n'
                    'rehearsals on a box with fewer GPUs than ranks)')
    args = ap.parse_args()

    # stdout must carry exactly one JSON line: native libraries (RCCL's banner, gloo) write to fd 1 too, so fd 1
    # points at stderr until the line is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    dist = None
    torch = None
    if args.gpus != world:
        print('bench.py: --gpus %d but WORLD_SIZE=%d; N > 1 must be launched through torch.distributed.run '
              '(one rank per GPU) -- running with %d rank(s)' % (args.gpus, world, world), file=sys.stderr)
    n_dev = 1
    force_dist = bool(os.environ.get('BLUEICE_BENCH_FORCE_DIST'))      # rehearse the N > 1 code path with one rank
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        n_dev = max(torch.cuda.device_count(), 1)
        if args.backend == 'nccl':
            torch.cuda.set_device(local_rank % n_dev)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank % n_dev))
        else:
            dist.init_process_group(args.backend)
    multi = world > 1 or force_dist
    use_cuda_tensors = multi and args.backend == 'nccl'

    from blueice_amd.device import DeviceContext
    from blueice_amd.synthetic import SyntheticModel

    K, W = args.steps, args.warmup
    model = SyntheticModel.named(args.config)
    ctx = DeviceContext(local_rank % n_dev if multi else local_rank)
    info = ctx.info()
    model.upload(ctx)
    counts = model.counts()
    ctx.set_param('sparse', 0)          # headline = the dense kernel: every evaluation visits every bin
    ctx.upload_counts(counts)

    # a pool of plans: host-side preparation done, descriptors resident on the device
    sets = [model.disjoint_cell_points(parity=i, seed=1000 * rank + i) for i in range(POOL)]
    plans = [ctx.plan(zz, rr) for zz, rr in sets]
    PPS = plans[0].P                                # points (evaluations) per step
    z, r = sets[0]
    bytes_per_launch = plans[0].bytes
    assert plans[0].launches == 1 and bytes_per_launch == PPS * 8 * (8 * model.S + 1) * model.B

    out_ptr = None
    if use_cuda_tensors:
        # results land directly in a torch (RCCL-visible) device tensor: no host round trip before the gather
        out = torch.empty(K * PPS, dtype=torch.float64, device='cuda')
        gathered = [torch.empty(K * PPS, dtype=torch.float64, device='cuda') for _ in range(world)]
        out_ptr = out.data_ptr()
    elif multi:
        out = torch.empty(K * PPS, dtype=torch.float64)
        gathered = [torch.empty(K * PPS, dtype=torch.float64) for _ in range(world)]

    def barrier():
        ctx.sync()
        if multi:
            if use_cuda_tensors:
                torch.cuda.synchronize()
            dist.barrier()
            if use_cuda_tensors:
                torch.cuda.synchronize()

    def run_steps(n, base=0):
        if out_ptr is not None:
            for i in range(n):
                plans[i % POOL].run(out_ptr + 8 * PPS * ((base + i) % K))
        else:
            for i in range(n):
                plans[i % POOL].run()

    run_steps(W)
    if multi:
        dist.all_gather(gathered, out)         # warm-up of the collective as well (communicator, channels, kernels)
    barrier()
    t0 = time.perf_counter()
    run_steps(K)
    ctx.sync()
    if multi:
        if not use_cuda_tensors:               # rehearsal backend: last step's results via the host
            last, _ = plans[(K - 1) % POOL].read()
            out[-PPS:] = torch.from_numpy(last)
        dist.all_gather(gathered, out)         # the final gather: the only collective (RCCL over xGMI)
        if use_cuda_tensors:
            torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if use_cuda_tensors else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank evaluated different points: the gathered vector must be finite everywhere
        tail = torch.stack([g[-PPS:] for g in gathered]).cpu().numpy()
        assert np.all(np.isfinite(tail)), 'gathered results contain non-finite values'

    # kernel time of the same steps, HIP events on the context stream around every launch
    ctx.profile(True)
    n_prof = min(K, 256)
    run_steps(n_prof)
    launches, ms = ctx.profile_read()
    ctx.profile(False)
    achieved = bytes_per_launch * launches / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

    # HBM bytes per launch from the PMC counters (rocprofv3 cannot run inside this process): taken from the
    # committed summary of the same command (profiles/), corrected as MI355X_MICROARCH.md prescribes
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')) as f:
            pm = json.load(f)
        if args.config == 'C2' and pm.get('algorithmic_bytes_per_launch') == bytes_per_launch:
            traffic = pm['traffic_bytes_per_launch']
    except Exception:
        traffic = None

    # the device's read-only streaming ceiling, measured live: a plain sum over the same resident tensor
    stream_ceiling = copy_ceiling = None
    if rank == 0:
        try:
            copy_ceiling = ctx.copy_bandwidth(1 << 31, reps=3)
            stream_ceiling = max(ctx.read_bandwidth(nontemporal=True, blocks_per_cu=b, reps=3) for b in (16, 32))
        except Exception as e:                       # a measurement aid only: never fail the bench line over it
            print('bench.py: read-bandwidth probe failed: %s' % e, file=sys.stderr)

    result = None
    if rank == 0:
        result = {
            'metric': 'likelihood evals/sec (and GB/s vs HBM peak), 4-src 5^3-anchor 100^3-bin model',
            'value': world * K * PPS / elapsed, 'unit': 'evals/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'C2: 4 sources, 3 shape params (5^3 anchors), 100^3 bins, single dataset; '
                                   'step = one batched call of %d independent dense evaluations in grid cells '
                                   'that share no anchor (no template re-use), tensor replicated per GPU' % PPS,
                       'sources': model.S, 'anchors': list(model.n_anchor), 'bins': list(model.bins),
                       'evals_per_step': PPS, 'device': info['arch']},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_source': 'profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                           'bytes per launch)' if traffic else None,
                         'kernel': 'k_morph_reduce<1,false,true> (G=1, no BB, nontemporal loads)', 'bytes_per_launch': bytes_per_launch,
                         'avg_launch_us': ms / max(launches, 1) * 1e3,
                         'stream_ceiling': stream_ceiling, 'copy_ceiling': copy_ceiling,
                         'stream_ceiling_note': 'GB/s of a plain 16-byte-load sum over the resident 4 GB tensor '
                                                '(nontemporal loads, best of 16 / 32 blocks per CU); copy_ceiling: bytes read + written '
                                                'per second of a 2 GiB device-to-device hipMemcpy; same process'},
        }

    if rank == 0 and world == 1 and not args.no_extras:
        ex = {}
        bytes_per_eval = bytes_per_launch // PPS
        zz, rr = model.stratified_points(seed=3)           # all 64 cells in one call: neighbours share corners
        p = ctx.plan(zz, rr)
        p.run()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(20):
            p.run()
        ctx.sync()
        ex['all_64_cells_batch_evals_per_s'] = 20 * len(zz) / (time.perf_counter() - t)
        p.close()
        singles = [ctx.plan(z[i], r[i]) for i in range(PPS)]     # one evaluation per launch, rotating cells
        for p in singles:
            p.run()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(8):
            for p in singles:
                p.run()
        ctx.sync()
        ex['one_point_per_launch_evals_per_s'] = 8 * PPS / (time.perf_counter() - t)
        for p in singles:
            p.close()
        p = ctx.plan(z[0], r[0])                           # same cell every call (a fit's access pattern)
        for _ in range(50):
            p.run()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(1000):
            p.run()
        ctx.sync()
        dt = (time.perf_counter() - t) / 1000
        ex['same_cell_evals_per_s'] = 1 / dt
        ex['same_cell_GBps'] = bytes_per_eval / dt / 1e9
        p.close()
        t = time.perf_counter()                            # full synchronous call incl. host planning + D2H
        for i in range(200):
            ctx.eval(z[i % PPS], r[i % PPS])
        ex['sync_call_latency_us'] = (time.perf_counter() - t) / 200 * 1e6
        zz, rr = model.random_points(16384, seed=7)        # scan batch: cell-grouped, templates reused
        p = ctx.plan(zz, rr)
        p.run()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(3):
            p.run()
        ctx.sync()
        ex['scan_batch_16384_evals_per_s'] = 3 * 16384 / (time.perf_counter() - t)
        p.close()
        zz, rr = model.random_points(131072, seed=11)      # the same on a scan of 131 072 points (128 items per cell)
        p = ctx.plan(zz, rr)
        p.run()
        ctx.sync()
        t = time.perf_counter()
        p.run()
        ctx.sync()
        ex['dense_scan_131072_evals_per_s'] = len(zz) / (time.perf_counter() - t)
        p.close()
        T = 256                                            # toy-MC: one point, T datasets (fp64 counts)
        toys = np.stack([model.counts(dataset=i) for i in range(T)])
        for mode, key in ((0, 'toy_mc_256_dense_counts_evals_per_s_kernels'), (1, 'toy_mc_256_csr_evals_per_s_kernels')):
            ctx.set_param('sparse', mode)
            ctx.upload_counts(toys)
            ctx.eval_datasets(z[0], r[0])
            ctx.profile(True)
            ctx.eval_datasets(z[0], r[0])
            _, tms = ctx.profile_read()
            ctx.profile(False)
            ex[key] = T / (tms * 1e-3)
        # non-empty-bin form (exact: templates >= 0): only the ~1e4 bins with data are visited per evaluation
        ctx.set_param('sparse', 1)
        ctx.upload_counts(counts)
        zz, rr = model.random_points(131072, seed=11)
        t = time.perf_counter()
        p = ctx.plan(zz, rr)
        t_plan = time.perf_counter() - t
        p.run()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(3):
            p.run()
        ctx.sync()
        dt = (time.perf_counter() - t) / 3
        ex['sparse_scan_131072_evals_per_s_device'] = len(zz) / dt
        ex['sparse_scan_131072_evals_per_s_incl_planning'] = len(zz) / (dt + t_plan)
        ex['sparse_nonempty_bins'] = ctx.get_param('nnz_total')
        p.close()
        # BASELINE.json configs[2] at full scale: 10^4 toy datasets drawn on the device at one parameter point,
        # all evaluated by one call at a nearby point (wall time includes the D2H of the 10^4 results)
        t = time.perf_counter()
        ctx.generate_toys(z[0], r[0], 10000, seed=1)
        ex['toy_mc_10000_generate_s'] = time.perf_counter() - t
        z_near = np.clip(z[0] + 0.03, [g[0] for g in model.anchor_z], [g[-1] for g in model.anchor_z])
        ctx.eval_datasets(z_near, r[0])
        t = time.perf_counter()
        for _ in range(5):
            ctx.eval_datasets(z_near, r[0])
        ex['toy_mc_10000_evals_per_s_wall'] = 5 * 10000 / (time.perf_counter() - t)
        ctx.set_param('sparse', 0)
        ctx.upload_counts(counts)
        # the same model end to end through the reference's API: Source plug-ins -> BinnedLogLikelihood.prepare()
        # -> set data -> inference.bestfit_scipy (first rate + the three shape parameters floating)
        t = time.perf_counter()
        lf = model.likelihood(device=ctx.device)
        ex['api_prepare_s'] = time.perf_counter() - t
        lf.set_binned_data(counts.reshape(model.bins))
        fixed = {'s%d_rate_multiplier' % s: 1 for s in range(1, model.S)}
        lf.bestfit_scipy(**fixed)
        t = time.perf_counter()
        best, ll = lf.bestfit_scipy(**fixed)
        ex['api_bestfit_scipy_s'] = time.perf_counter() - t
        t = time.perf_counter()
        lf.bestfit_scipy(use_gradient=True, **fixed)
        ex['api_bestfit_scipy_with_gradient_s'] = time.perf_counter() - t
        t = time.perf_counter()
        for i in range(300):
            lf(shape0=0.1 + 1e-4 * i, s0_rate_multiplier=1.05)
        ex['api_call_us'] = (time.perf_counter() - t) / 300 * 1e6
        ex['api_bestfit_max_loglikelihood'] = ll
        # BASELINE.json configs[3] on one GPU: a profile scan of 10^6 parameter points through lf.eval_points
        g = np.random.default_rng(5)
        pts = dict(shape0=g.uniform(-2, 2, 10 ** 6), shape1=g.uniform(-2, 2, 10 ** 6), s0_rate_multiplier=g.uniform(0.8, 1.2, 10 ** 6))
        lf.eval_points(pts)
        t = time.perf_counter()
        lf.eval_points(pts)
        ex['api_eval_points_1e6_s'] = time.perf_counter() - t
        del lf
        result['extras'] = ex
        # BASELINE.json's north star names two targets; where each one is met
        result['north_star'] = {
            'hbm_frac_target': 0.70, 'hbm_frac': result['roofline']['frac'],
            'evals_per_s_target': 1e6,
            'dense_evals_per_s_ceiling_no_reuse': HBM_PEAK_GBS * 1e9 / bytes_per_eval,
            'evals_per_s_scan_every_bin_visited': ex['dense_scan_131072_evals_per_s'],
            'evals_per_s_scan_default_path_incl_planning': ex['sparse_scan_131072_evals_per_s_incl_planning'],
            'note': 'an evaluation that shares no template bytes with its neighbours moves %.0f MB, so 8 TB/s caps it at '
                    '%.1f k/s: `value` is that case, at `roofline.frac` of the peak.  10^6/s needs re-use: a scan '
                    'of 131072 points reads each cell\'s rows once (matrix-core kernel, every bin visited), and the '
                    'default path (exact non-empty-bin identity, templates >= 0) visits only the %d bins with data'
                    % (bytes_per_eval / 1e6, HBM_PEAK_GBS * 1e9 / bytes_per_eval / 1e3, ex['sparse_nonempty_bins']),
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(model, counts, (z, r))
        result['cpu_baseline']['host_cores_available'] = os.cpu_count()
        try:
            n_procs = max(1, min(16, os.cpu_count() or 1))          # a 1-GPU box's CPU share
            result['cpu_baseline']['all_cores'] = cpu_baseline_all_cores(args.config, n_procs)
        except Exception as e:                                       # never let the side figure break the line
            result['cpu_baseline']['all_cores'] = {'error': repr(e)}
    elif rank == 0:
        result['cpu_baseline'] = None

    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)
    for p in plans:
        p.close()
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
