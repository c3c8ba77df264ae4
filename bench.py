#!/usr/bin/env python3
"""Headline benchmark: binned-likelihood evaluations per second on the BASELINE.json model.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C4-dense|C5|C5-2anchor]

Default workload (BASELINE.json configs[1], "C2"): 4 sources, 3 shape parameters with 5 anchors each (125 anchor
models), 100x100x100 analysis bins; the anchor tensor (4.0 GB fp64) and the data are resident in HBM before the
timed region.  One STEP = one batched call evaluating 8 independent parameter points, each against a dataset of its
own: for every point the morph+reduce kernel streams the 2^3 * 4 corner templates of its grid cell plus its counts
row (264 MB per evaluation) and reduces to a scalar -- 2.1 GB per step, ONE kernel launch (the
last block of every evaluation finishes it inside the launch).  The 8 points of a step lie in grid cells that share
no anchor model with each other and use 8 different datasets, so no byte is used twice within a step and a step's
2.1 GB is far beyond the 256 MiB Infinity Cache: algorithmic bytes = compulsory HBM traffic.  Successive steps
rotate through the 8 parity combinations of cells.

With N > 1 ranks (one process per GPU) every rank holds a replica.  `python bench.py --gpus N` with no WORLD_SIZE in the
environment starts the N rank processes ITSELF: the parent -- before it has touched the GPU or loaded the library --
spawns N fresh children of this script (blueice_amd.launch: RANK / LOCAL_RANK / WORLD_SIZE / rendezvous file in their
environment), relays rank 0's JSON line (the children inherit its stdout) and exits with the first non-zero child code.
Launched by torch.distributed.run or `python -m blueice_amd.launch` only the RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*
environment is read (PyTorch is not imported).  Every rank holds a replica
of the tensor and evaluates its own K steps (weak scaling, no data-path collective); the per-rank result vectors stay
in HBM and are gathered once at the end with RCCL (ncclAllGather, bound directly: blueice_amd.comm), inside the timed
region.  The configurations that really shard -- 10^6 scan points dealt by grid cell (configs[3]; the dealing is the
device planner's sort, inside the timed step), 10^4 toy datasets split by range (configs[2]), 256 Beeston-Barlow scan
points on a grid cell of configs[4] -- run as STRONG-scaling legs behind the headline on every N and are reported under
"legs" (fixed total work, gather inside the timed region, cross-rank consistency asserted on a sample).

The JSON line also carries
  roofline      morph+reduce kernel: algorithmic bytes per launch / HIP-event kernel time vs 8 TB/s, the measured
                stream ceiling of the same access pattern, PMC traffic from profiles/
  cpu_baseline  the numpy/scipy oracle (the reference's arithmetic) timed on the host, rank 0, N = 1
  legs          C4 / C4-dense / C3 / C5-BB (every N; C5-BB carries the Beeston-Barlow kernel's own roofline); unbinned (N = 1)
  extras        other call shapes of the same path (N = 1)
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
FP64_PEAK_TFLOPS = 78.6     # fp64 matrix = fp64 vector peak (MI355X_MICROARCH.md)
POOL = 8
METRIC = 'likelihood evals/sec (and GB/s vs HBM peak), 4-src 5^3-anchor 100^3-bin model'


def log(*a):
    print(*a, file=sys.stderr, flush=True)


HOST_THREADS_MAX = 16        # the library's own limit (csrc/bi_planning.h: host_threads), and a one-GPU box's CPU share


def host_thread_share(world):
    """Host threads ONE rank may use: the cores this process may run on, divided among the ranks of the node (every rank
    uploads its replica and plans its batches with a few threads; eight ranks x 16 threads would be a worker pool the GPU
    pool kills)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return max(1, min(HOST_THREADS_MAX, cores // max(1, int(world))))


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (the only place the oracle is used)
# ---------------------------------------------------------------------------------------------------------
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


def cpu_baseline(model, counts, points, budget_s=15.0):
    """Time the oracle (numpy/scipy restatement of the reference path) on this host, one thread."""
    from oracle import blueice_oracle as orc
    z, r = points
    cm = model.cell_model(z[0])
    value = float(orc.loglikelihood(cm, counts, z[0], r[0]))            # warm; also the checker of the timed outputs
    n, t0 = 0, time.perf_counter()
    while True:
        orc.loglikelihood(cm, counts, z[0], r[0])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 2000:
            break
    return dict(value=n / dt, unit='evals/s', cores=1, kind='port', oracle_value=value,
                sample='%d single-thread evaluations of the C2 model at one off-grid point (%.1f s); '
                       'numpy %s oracle = the reference arithmetic' % (n, dt, np.__version__))


# ---------------------------------------------------------------------------------------------------------
# communicator plumbing
# ---------------------------------------------------------------------------------------------------------
class Ranks:
    """World of this run + the gather buffers in HBM."""

    def __init__(self, ctx, backend):
        # --backend given explicitly: that gather or none (the run ends non-zero if a fallback had to be taken); not given:
        # RCCL, with the agreed socket fallback allowed -- a gather of a few MB must not be what fails an unattended run
        self.strict = backend is not None
        backend = backend or 'rccl'
        self.requested = backend
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.rank = int(os.environ.get('RANK', 0))
        self.ctx = ctx
        self.comm = None
        self.multi = self.world > 1 or bool(os.environ.get('BLUEICE_BENCH_FORCE_DIST'))
        if self.multi:
            from blueice_amd.comm import CommInitTimeout, connect
            t = time.perf_counter()
            try:
                self.comm = connect(ctx, backend=backend, rank=self.rank, world=self.world)
            except CommInitTimeout as e:
                # ncclCommInitRank never came back on some rank: this run is over.  A rank whose init thread is still
                # inside RCCL cannot shut down in an orderly way -- leave at once, non-zero (the launcher ends the others)
                log('rank %d/%d: %s -- giving up' % (self.rank, self.world, e))
                sys.stderr.flush()
                os._exit(3)
            log('rank %d/%d: communicator %s ready in %.1f s %s' % (
                self.rank, self.world, self.comm.kind, time.perf_counter() - t, getattr(self.comm, 'fallback_reason', '')))
        self.device_gather = self.comm is not None and hasattr(self.comm, 'all_gather_device')
        self._send = self._recv = self._full = None

    @property
    def kind(self):
        if self.comm is None:
            return 'none (one process)'
        if self.comm.kind == 'rccl':
            return 'rccl %s, ncclAllGather on the context stream between device buffers (ctypes binding)' % self.comm.version
        return '%s (host)%s' % (self.comm.kind, ' -- RCCL unavailable: ' + self.comm.fallback_reason
                                if getattr(self.comm, 'fallback_reason', '') else '')

    def report(self):
        """What the JSON line says about the gather (collective: every rank calls it)."""
        from blueice_amd.comm import describe
        # backend_explicit: --backend was given (a fallback then ends the run non-zero); else RCCL is preferred and a fallback allowed
        if self.comm is None:
            return dict(backend_requested=self.requested, backend_explicit=self.strict, gather_kind='none (one process)', rccl_ranks=None,
                        rccl_version=None, rank_devices=[self.ctx.device], rank_devices_source='this process', gather_fallback_reason=None)
        return dict(describe(self.comm, self.requested, local_device=self.ctx.device), backend_explicit=self.strict)

    def fallback_taken(self):
        return self.strict and self.comm is not None and self.requested == 'rccl' and self.comm.kind != 'rccl'

    def buffers(self, n):
        """send [n] and recv [world * n] doubles in HBM."""
        if self._send is None or self._send.nbytes < 8 * n:
            for b in (self._send, self._recv):
                if b is not None:
                    b.free()
            self._send = self.ctx.device_alloc(8 * n)
            self._recv = self.ctx.device_alloc(8 * n * max(self.world, 1))
        return self._send, self._recv

    def gather(self, n):
        """All ranks' send[0:n] -> host array [world, n]; the send buffer must have been filled on the context stream."""
        send, recv = self.buffers(n)
        if self.device_gather:
            self.comm.all_gather_device(send.ptr, recv.ptr, n)
            return recv.to_host(np.float64, n * self.world).reshape(self.world, n)
        local = send.to_host(np.float64, n)
        if self.comm is None:
            return local[None, :]
        return self.comm.all_gather(local)

    def gather_on_device(self, n):
        """All ranks' send[0:n] -> recv [world, n], left in HBM (RCCL on the context stream; or through the host communicator)."""
        send, recv = self.buffers(n)
        if self.device_gather:
            self.comm.all_gather_device(send.ptr, recv.ptr, n)
        elif self.comm is not None:
            recv.from_host(self.comm.all_gather(send.to_host(np.float64, n)))
        else:
            recv.from_host(send.to_host(np.float64, n))
        return recv

    def full_buffer(self, n):
        if self._full is None or self._full.nbytes < 8 * n:
            if self._full is not None:
                self._full.free()
            self._full = self.ctx.device_alloc(8 * n)
        return self._full

    def agree_status(self, word, what):
        """OR of a status word over the ranks; a launch that gave up on a partial sum anywhere fails the leg everywhere."""
        if self.comm is not None:
            word = int(self.comm.all_reduce(np.array([word], dtype=np.int64), 'bor')[0])
        assert not word & 32, '%s: a launch gave up waiting for a partial sum (BI_ST_INTERNAL)' % what
        return word

    def barrier(self):
        self.ctx.sync()
        if self.comm is not None:
            self.comm.barrier()
            self.ctx.sync()

    def max_over_ranks(self, x):
        if self.comm is None:
            return float(x)
        return float(self.comm.all_reduce(np.array([float(x)]), 'max')[0])

    def close(self):
        for b in (self._send, self._recv, self._full):
            if b is not None:
                b.free()
        self._send = self._recv = self._full = None
        if self.comm is not None:
            self.comm.close()


# ---------------------------------------------------------------------------------------------------------
# strong-scaling legs
# ---------------------------------------------------------------------------------------------------------
def predicted_ceiling(units, step_ms, kernel_ms, world, undivided_kernel_ms=0.0):
    """What a strong-scaling leg can reach at N ranks, from this run's own split of a step: the kernels' share divides by
    N, everything else (planning of ALL units on every rank, copies, the gather, the read-back) is paid by every rank in
    full.  `kernel_ms` is this run's kernel time per step on the busiest rank (HIP events), i.e. 1/world of the total;
    `undivided_kernel_ms` is the part of it that every rank repeats in full (the toy leg's log mu pass over all bins)."""
    kernel_ms = min(kernel_ms, step_ms)          # (the step that carries the HIP events is a little slower than the timed ones)
    undivided = min(max(0.0, undivided_kernel_ms), kernel_ms)
    fixed = step_ms - kernel_ms + undivided
    total_kernel = (kernel_ms - undivided) * world
    return dict(model='per-rank fixed ms + kernel ms / N', fixed_ms_per_rank=fixed, kernel_ms_total=total_kernel,
                kernel_ms_repeated_by_every_rank=undivided, measured_at_n=world,
                evals_per_s={str(n): units / ((fixed + total_kernel / n) * 1e-3) for n in (1, 2, 4, 8)})


def scan_leg(ctx, ranks, model, P, steps, label, sample=4):
    """A profile scan of P parameter points over the resident model (BASELINE.json configs[3]), STRONG scaling.  One step =
    ALL P points in -> full result vector on every rank, everything inside the clock.  Timed twice: with the points
    handed over as host arrays (`value_host_points`: the reference's calling convention, H2D inside the clock) and with
    the points already in HBM when the clock starts (`value`; bi_plan_points_resident).  Per step:
      N > 1  every rank hands all P points to its device planner, whose (cell, dataset) sort IS the dealing: rank r
             evaluates a contiguous, balanced range of the sorted list (cells stay together, no host pass over the
             points); the ranks' vectors are gathered in HBM (RCCL) and one kernel scatters them into point order;
      N = 1  plan, evaluate, one copy back.
    Beeston-Barlow models are planned (and dealt) on the device as well since round 4 (work items of bb_max_group points).
    Replaces the reference's Python double loop over lf(**kw) (blueice/inference.py:424-432)."""
    from blueice_amd.sharding import deal_points_by_cell
    world, rank = ranks.world, ranks.rank
    work = [model.random_points(P, seed=900 + s) for s in range(steps + 1)]      # step 0 is the warm-up
    device_deal = world > 1
    can_reside = True
    # The device planner refuses (ValueError: BI_ERR_INVALID) Beeston-Barlow batches in which some bin can have U_b == 0, or
    # with bb_exact = 1, and infinite rates of a source that may go negative: such a leg is dealt on the host and planned
    # from host arrays, on every rank alike (the ranks agree before the first step)
    refused = 0
    try:
        probe = ctx.plan_share(work[0][0], work[0][1], None, rank, world) if device_deal else None
        if probe is None:
            bz, br = ctx.device_alloc(work[0][0].nbytes), ctx.device_alloc(work[0][1].nbytes)
            bz.from_host(work[0][0]); br.from_host(work[0][1])
            try:
                probe = ctx.plan_resident(P, bz, br)
            finally:
                bz.free(); br.free()
        probe.close()
    except ValueError as e:
        log('%s: the device planner refuses this batch (%s): host dealing, host-array plans' % (label, e))
        refused = 1
    if ranks.comm is not None:
        refused = int(ranks.comm.all_reduce(np.array([refused], dtype=np.int64), 'bor')[0])
    if refused:
        device_deal, can_reside = False, False
    stride = P if world == 1 else -(-P // world)
    send, _ = ranks.buffers(stride)
    send.from_host(np.zeros(stride))
    full = ranks.full_buffer(P) if device_deal else None
    seen = dict(bytes=0, share=[P, P])
    host_results = np.empty(P)                        # (reused: a fresh 8 MB array per step costs page faults and the copy's slow path)

    def step(w, resident=None, fetch=True):
        """fetch = False: the full result vector stays in HBM (where a device-side consumer -- the batched fit engine, a reduction
        -- reads it); only the plan's status word comes back.  The caller's own copy is one more D2H of 8 P bytes."""
        z, r = w
        if device_deal:
            # (host points: H2D of the points first)  geometry, sort, this rank's items
            plan = ctx.plan_resident(P, resident[0], resident[1], None, rank, world) if resident else ctx.plan_share(z, r, None, rank, world)
            plan.run(send.ptr)
            word = plan.status()
            recv = ranks.gather_on_device(stride)
            plan.unsort(recv.ptr, stride, full.ptr)
            out = full.to_host(np.float64, P, out=host_results) if fetch else None
            seen['share'] = [plan.n_valid // world, -(-plan.n_valid // world)]
        elif world == 1:
            plan = ctx.plan_resident(P, resident[0], resident[1]) if resident else ctx.plan(z, r)
            plan.run(send.ptr)
            word = plan.status()
            out = send.to_host(np.float64, P, out=host_results) if fetch else None
        else:
            deal = deal_points_by_cell(model.anchor_z, z, world)
            mine = deal[rank]
            plan = ctx.plan(z[mine], r[mine]) if len(mine) else None
            word = 0
            if plan is not None:
                plan.run(send.ptr)
                word = plan.status()
            parts = ranks.gather(stride)
            out = np.empty(P)
            for idx, vals in zip(deal, parts):
                out[idx] = vals[:len(idx)]
            seen['share'] = [min(len(d) for d in deal), max(len(d) for d in deal)]
        if plan is not None:
            seen['bytes'] = plan.bytes                # bytes the launches of this rank's share stream (work items x rows)
            plan.close()
        ranks.agree_status(word, label)
        return out

    step(work[0])
    ranks.barrier()
    t0 = time.perf_counter()
    for w in work[1:]:
        out = step(w)
    ranks.barrier()
    elapsed_host = ranks.max_over_ranks(time.perf_counter() - t0)
    out = out.copy()                                      # (the steps reuse one host array)
    # the same steps with the points already in HBM when the clock starts (bi_plan_points_resident): the leg's `value`;
    # the rate with the points handed over as host arrays -- the reference's calling convention, H2D inside -- beside it
    elapsed = elapsed_host
    if can_reside:
        held = []
        for z, r in work:
            bz, br = ctx.device_alloc(z.nbytes), ctx.device_alloc(r.nbytes)
            bz.from_host(z); br.from_host(r)
            held.append((bz, br))
        for _ in range(3):                                # (the chip needs a few steps after the uploads above: 11.0, 10.6, 10.1, 9.9 ms)
            step(work[0], held[0])
        ranks.barrier()
        t0 = time.perf_counter()
        for w, h in zip(work[1:], held[1:]):
            out_res = step(w, h)
        ranks.barrier()
        elapsed_fetch = ranks.max_over_ranks(time.perf_counter() - t0)
        out_res = out_res.copy()
        # ... and with the full vector LEFT in HBM on every rank (status word checked): inputs and outputs on the device, the
        # form a device-side consumer of a scan sees -- the leg's `value`; the rate with the caller's own host copy beside it
        step(work[0], held[0], fetch=False)
        ranks.barrier()
        t0 = time.perf_counter()
        for w, h in zip(work[1:], held[1:]):
            step(w, h, fetch=False)
        ranks.barrier()
        elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
        kept = (full if device_deal else send).to_host(np.float64, P)
        assert np.array_equal(kept, out_res), '%s: the vector left in HBM differs from the one fetched' % label
        if ctx.bb_source >= 0:      # (small Beeston-Barlow batches of host arrays are planned on the host: other item sizes, other block counts)
            assert np.allclose(out_res, out, rtol=1e-12, atol=0), '%s: resident points and host points disagree' % label
        else:
            assert np.array_equal(out_res, out), '%s: resident points and host points disagree' % label
    # where a step's time goes: one more step with HIP events around every kernel launch of this rank
    # (short steps three times over: one step's kernel times scatter by a per cent or two)
    n_prof = 3 if elapsed / steps < 0.05 else 1
    ctx.profile(True)
    for _ in range(n_prof):
        step(work[-1], held[-1] if can_reside else None, fetch=not can_reside)
    _, kernel_ms = ctx.profile_read()
    ctx.profile(False)
    kernel_ms = ranks.max_over_ranks(kernel_ms / n_prof)
    if can_reside:
        for bz, br in held:
            bz.free(); br.free()
    # What ONE rank of an N-rank run does per step, measured here (one process, one GPU): this rank plans and evaluates share N/2 of
    # N of the same resident points (bi_plan_points_resident with share_world = N) and reads the status word -- everything of a
    # rank's step except the gather of the N vectors and the scatter into point order.  A direct measurement of what the
    # predicted_ceiling model guesses from the N = 1 split (the model takes the whole N = 1 "everything else" as fixed per rank).
    rehearsed = None
    if world == 1 and can_reside and not refused:
        rehearsed = {}
        bz, br = ctx.device_alloc(work[-1][0].nbytes), ctx.device_alloc(work[-1][1].nbytes)
        bz.from_host(work[-1][0]); br.from_host(work[-1][1])
        for n_ranks in (2, 4, 8):
            def share_step():
                plan = ctx.plan_resident(P, bz, br, None, n_ranks // 2, n_ranks)
                plan.run(send.ptr)
                word = plan.status()
                plan.close()
                return word
            share_step()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                share_step()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            rehearsed[str(n_ranks)] = dict(ms_per_step_of_one_share_without_gather=ms, evals_per_s_if_every_rank_does_alike=P / (ms * 1e-3))
        bz.free(); br.free()
    # consistency on a sample, through a DIFFERENT kernel path (the single-point kernel) on THIS rank: every rank holds
    # the whole tensor, so any rank can check any point, whoever evaluated it
    z, r = work[-1]
    picks = np.random.default_rng(17 + rank).choice(P, size=min(P, 2 * sample), replace=False)
    worst = 0.0
    for i in picks:
        one, _ = ctx.eval(z[i], r[i])
        worst = max(worst, abs(one[0] - out[i]) / max(1.0, abs(out[i])))
    assert worst <= 1e-11, '%s: gathered scan differs from single evaluations by %.2e' % (label, worst)
    assert np.all(np.isfinite(out)), '%s: non-finite values in the gathered scan' % label
    return dict(workload=label, scaling='strong', points=P, steps=steps, value=P * steps / elapsed, unit='evals/s',
                ms_per_step=elapsed / steps * 1e3,
                inputs=('points resident in HBM when the clock starts (bi_plan_points_resident); the full result vector is left in HBM on '
                        'every rank, its status word read back' if can_reside
                        else 'host arrays (the device planner refused the batch: host planner, H2D inside the clock)'),
                value_results_to_host=(P * steps / elapsed_fetch) if can_reside else None,
                ms_per_step_results_to_host=(elapsed_fetch / steps * 1e3) if can_reside else None,
                value_host_points=P * steps / elapsed_host, ms_per_step_host_points=elapsed_host / steps * 1e3,
                points_per_rank_min_max=seen['share'],
                dealing=('device planner sort, inside the step' if device_deal else
                         ('none (one process)' if world == 1 else 'host (deal_points_by_cell), inside the step')),
                sample_max_rel_diff_vs_single_point_kernel=worst, streamed_bytes_this_rank=int(seen['bytes']), gather=ranks.kind,
                step_split_ms=dict(kernels_busiest_rank=kernel_ms, everything_else=max(0.0, elapsed / steps * 1e3 - kernel_ms)),
                predicted_ceiling=predicted_ceiling(P, elapsed / steps * 1e3, kernel_ms, world),
                rehearsed_share=rehearsed)


def toy_leg(ctx, ranks, model, T, steps):
    """BASELINE.json configs[2]: T toy-MC datasets, one parameter point per call, STRONG scaling: every rank draws its
    range of the SAME ensemble on the device (toy_offset: the Philox counters are global dataset numbers) and evaluates
    it; the T/world results stay in HBM and are gathered per call.  Replaces the user's loop over
    base_model.simulate() -> set_data -> lf() (blueice/model.py:69-91)."""
    from blueice_amd.sharding import split_range
    world, rank = ranks.world, ranks.rank
    spans = [split_range(T, q, world) for q in range(world)]
    t0_, t1_ = spans[rank]
    z, r = model.default_point()
    lo = np.array([g[0] for g in model.anchor_z])
    hi = np.array([g[-1] for g in model.anchor_z])
    ctx.set_param('sparse', 1)
    ctx.set_param('toy_offset', t0_)
    tg = time.perf_counter()
    ctx.generate_toys(z, r, t1_ - t0_, seed=4242)
    gen_s = ranks.max_over_ranks(time.perf_counter() - tg)
    nnz = int(ctx.get_param('nnz_total'))              # non-empty bins over this rank's datasets
    n_max = max(b - a for a, b in spans)
    send, _ = ranks.buffers(n_max)
    send.from_host(np.zeros(n_max))

    # (the parameter points of the timed calls are inputs: made before the clock starts)
    points = {k: np.ascontiguousarray(np.clip(z + 0.01 * (k + 1), lo, hi)) for k in range(-1, steps)}

    def step(k):
        zk = points[k]
        if world == 1:
            # one process: the finish kernel writes the T results straight into pinned host memory (bi_eval_datasets) -- the
            # "gather" of one rank is that write; N > 1 leaves them in HBM for the RCCL gather (bi_eval_datasets_device)
            out, st = ctx.eval_datasets(zk, r, 0, t1_ - t0_)
            return zk, st, out
        st = ctx.eval_datasets_device(send.ptr, zk, r, 0, t1_ - t0_)
        parts = ranks.gather(n_max)
        return zk, st, np.concatenate([p[:b - a] for p, (a, b) in zip(parts, spans)])

    step(-1)
    ranks.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        zk, st, out = step(k)
    ranks.barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    assert st == 0 and out.shape == (T,) and np.all(np.isfinite(out))
    # cross-rank consistency: re-draw single toys of OTHER ranks' ranges here and evaluate them (same counters ->
    # the same toy -> the same bits)
    checked = []
    for t in sorted({spans[(rank + 1) % world][0], spans[(rank + world - 1) % world][1] - 1, T // 2}):
        ctx.set_param('toy_offset', t)
        ctx.generate_toys(z, r, 64, seed=4242)         # 64: the batch size from which the tiled kernel takes the call, as above
        one, _ = ctx.eval_datasets(zk, r)
        assert one[0] == out[t], 'toy %d: %r on this rank, %r gathered' % (t, one[0], out[t])
        checked.append(t)
    ctx.set_param('toy_offset', t0_)
    ctx.generate_toys(z, r, t1_ - t0_, seed=4242)      # this rank's range again, for the kernel timing of one more call
    step(-1)
    ctx.profile(True)
    step(0)
    n_launch, kernel_ms = ctx.profile_read()
    ctx.profile(False)
    kernel_ms = ranks.max_over_ranks(kernel_ms)
    # the part of the kernels that does not shrink with the number of ranks: the log mu pass over all bins (a call on 64
    # datasets is that pass and next to nothing else)
    ctx.eval_datasets(points[0], r, 0, min(64, t1_ - t0_))
    ctx.profile(True)
    ctx.eval_datasets(points[0], r, 0, min(64, t1_ - t0_))
    _, logmu_ms = ctx.profile_read()
    ctx.profile(False)
    logmu_ms = ranks.max_over_ranks(logmu_ms)
    ctx.set_param('toy_offset', 0)
    # algorithmic bytes of a call on this rank: the 2^d*S template rows once (log mu of every bin) + one list entry (2 bytes: the
    # bin's byte offset within its tile and a count of at most 7 -- what these toys hold; 4 bytes otherwise) per non-empty bin of
    # every dataset; the entries are gathered against a tile of log mu held in LDS (runs are padded to 16-byte groups, not
    # counted here)
    NS = 2 ** model.d * model.S
    entry_bytes = int(ctx.get_param('tm_entry_bytes'))
    nbytes = 8.0 * NS * model.B + float(entry_bytes) * nnz
    step_ms = elapsed / steps * 1e3
    roof = dict(bound='hbm', unit='GB/s', peak=HBM_PEAK_GBS, bytes_per_call=nbytes, list_entry_bytes=entry_bytes, launches_per_call=int(n_launch),
                kernel_ms_per_call=kernel_ms, achieved=nbytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None,
                achieved_over_whole_step=nbytes / (step_ms * 1e-3) / 1e9, traffic=None,
                note='algorithmic bytes per call / summed kernel time of the call (HIP events): the log-mu pass streams the '
                     '2^d*S rows at HBM rate, the dataset pass streams the list entries (%d bytes each) against log mu tiles in '
                     'LDS; what the call is measured by is evaluations per second' % entry_bytes)
    roof['frac'] = roof['achieved'] / HBM_PEAK_GBS if roof['achieved'] else None
    return dict(workload='C3: 10^4 toy datasets (drawn on the device), one parameter point per call, datasets split '
                         'by range over the ranks', scaling='strong', datasets=T, steps=steps, value=T * steps / elapsed,
                unit='evals/s', ms_per_step=step_ms, generate_s=gen_s, toys_rechecked_bitwise=checked,
                nonempty_bins_this_rank=nnz, gather=ranks.kind, roofline=roof,
                step_split_ms=dict(kernels_busiest_rank=kernel_ms, everything_else=max(0.0, step_ms - kernel_ms)),
                predicted_ceiling=predicted_ceiling(T, step_ms, kernel_ms, world, undivided_kernel_ms=logmu_ms))


def toy_points_leg(ctx, ranks, model, T, P, steps):
    """BASELINE.json configs[2] the way toy-MC users run it: every simulated dataset evaluated at every hypothesis (the loops of
    blueice/inference.py:392-443 around blueice/model.py:69-91).  One step = P hypotheses x T datasets = P * T evaluations,
    STRONG scaling: every rank draws the WHOLE ensemble on the device (the Philox counters are global dataset numbers: the same
    toys on every rank), the hypotheses are dealt to the ranks by grid cell, a rank evaluates its P / N hypotheses with
    bi_eval_datasets_points (four hypotheses share a pass over the lists, the hypotheses of a cell the pass over its templates),
    leaves [P / N][T] in HBM, and the blocks are gathered per step.  Beside it, on this rank: the call with 1 and with 4
    hypotheses (evaluations per second per GPU)."""
    from blueice_amd.sharding import deal_points_by_cell
    world, rank = ranks.world, ranks.rank
    z0, r0 = model.default_point()
    lo = np.array([g[0] for g in model.anchor_z])
    hi = np.array([g[-1] for g in model.anchor_z])
    ctx.set_param('sparse', 1)
    ctx.set_param('toy_offset', 0)
    tg = time.perf_counter()
    ctx.generate_toys(z0, r0, T, seed=4242)
    gen_s = ranks.max_over_ranks(time.perf_counter() - tg)
    nnz = int(ctx.get_param('nnz_total'))

    def hypotheses(k, n=P):
        """n hypotheses of step k: n / 8 shape points (grid cells) x 8 signal strengths -- inputs, made before the clock starts"""
        shapes = max(1, n // 8)
        zs = np.stack([lo + np.mod(z0 - lo + 0.01 * (k + 1) + 1.05 * s, hi - lo) for s in range(shapes)])     # (different cells)
        z = np.repeat(zs, n // shapes, axis=0)
        r = np.repeat(r0[None, :], len(z), axis=0)
        r[:, 0] *= np.tile(np.linspace(0.25, 2.0, n // shapes), shapes)
        return np.ascontiguousarray(z), np.ascontiguousarray(r)

    work = {k: hypotheses(k) for k in range(-1, steps)}
    deals = {k: deal_points_by_cell(model.anchor_z, work[k][0], world) for k in work}
    n_max = max(max(len(d) for d in deal) for deal in deals.values())
    send, _ = ranks.buffers(n_max * T)
    send.from_host(np.zeros(n_max * T))

    result_buffer = np.empty((P, T))                     # (reused: a fresh 2.5 MB array per call costs ~0.1 ms of page faults)

    def step(k, fetch=True):
        """fetch = False: the [P][T] matrix stays in HBM (where a device-side consumer -- a reduction to test statistics, the next
        stage of a toy-MC chain -- takes it; one rank: this rank's send buffer, N ranks: the gathered [N][n_max T] buffer) and only
        the status words come back: the leg's `value`, like the scan legs'.  fetch = True: the matrix is also copied to the host."""
        z, r = work[k]
        if world == 1:
            if fetch:
                out, st = ctx.eval_datasets_points(z, r, out=result_buffer)
            else:
                out, st = None, ctx.eval_datasets_points_device(send.ptr, z, r)
            return out, int(np.bitwise_or.reduce(st))
        mine = deals[k][rank]
        st = ctx.eval_datasets_points_device(send.ptr, z[mine], r[mine]) if len(mine) else np.zeros(0, np.int32)
        out = None
        if fetch:
            parts = ranks.gather(n_max * T)
            out = np.empty((P, T))
            for idx, vals in zip(deals[k], parts):
                out[idx] = vals[:len(idx) * T].reshape(len(idx), T)
        else:
            ranks.gather_on_device(n_max * T)
            ctx.sync()
        return out, int(np.bitwise_or.reduce(st)) if len(st) else 0

    for _ in range(3):
        step(-1)
    ranks.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        out, st = step(k)
    ranks.barrier()
    elapsed_fetch = ranks.max_over_ranks(time.perf_counter() - t0)
    assert st == 0 and out.shape == (P, T) and np.all(np.isfinite(out))
    step(-1, fetch=False)
    ranks.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        _, st2 = step(k, fetch=False)
    ranks.barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    assert st2 == 0
    if world == 1:                                       # what was left in HBM is what the fetching step delivered (same call, same bits)
        kept = send.to_host(np.float64, P * T).reshape(P, T)
        assert np.array_equal(kept, out), 'toy hypotheses: the matrix left in HBM differs from the one fetched'
    else:
        kept = ranks.buffers(n_max * T)[1].to_host(np.float64, world * n_max * T).reshape(world, n_max * T)
        for idx, vals in zip(deals[steps - 1], kept):
            assert np.array_equal(vals[:len(idx) * T].reshape(len(idx), T), out[idx]), 'toy hypotheses: the gathered matrix left in HBM differs from the one fetched'
    # cross-rank consistency through ANOTHER path: hypotheses another rank evaluated, re-evaluated here one at a time
    # (bi_eval_datasets: tiles of 8192 bins, one point per pass) -- the same sums in another grouping
    z, r = work[steps - 1]
    worst = 0.0
    for p in sorted({0, P // 2 + 1, P - 1}):
        one, _ = ctx.eval_datasets(z[p], r[p])
        worst = max(worst, float(np.max(np.abs(one - out[p]) / np.abs(one))))
    assert worst <= 1e-12, 'toy hypotheses: gathered values differ from single-point calls by %.2e' % worst
    step(0, fetch=False)
    ctx.profile(True)
    for _ in range(3):
        step(0, fetch=False)
    n_launch, kernel_ms = ctx.profile_read()
    ctx.profile(False)
    n_launch, kernel_ms = n_launch // 3, ranks.max_over_ranks(kernel_ms / 3)
    step_ms = elapsed / steps * 1e3
    # the call by number of hypotheses, on this rank (per GPU): one point per call (bi_eval_datasets), four, all P
    per_gpu = {}
    for n_h in (1, 4, P):
        zh, rh = hypotheses(0, max(n_h, 8))
        zh, rh = zh[:n_h], rh[:n_h]                      # (n_h <= 8: signal strengths of one shape point -- one grid cell)
        fn = (lambda: ctx.eval_datasets(zh[0], rh[0])) if n_h == 1 else (lambda: ctx.eval_datasets_points(zh, rh))
        for _ in range(3):
            fn()
        reps = max(5, 60 // n_h)
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        per_gpu[str(n_h)] = n_h * T * reps / (time.perf_counter() - t)
    # what ONE rank of an N-rank run does per step, measured here: the hypotheses dealt to rank 0 of N, evaluated into HBM
    # (bi_eval_datasets_points_device) -- a rank's whole step except the gather
    rehearsed = None
    if world == 1:
        rehearsed = {}
        zr, rr = work[0]
        for n_ranks in (2, 4, 8):
            mine = deal_points_by_cell(model.anchor_z, zr, n_ranks)[0]
            if not len(mine):
                continue
            zm, rm = np.ascontiguousarray(zr[mine]), np.ascontiguousarray(rr[mine])
            for _ in range(2):
                ctx.eval_datasets_points_device(send.ptr, zm, rm)
            t = time.perf_counter()
            for _ in range(10):
                ctx.eval_datasets_points_device(send.ptr, zm, rm)
            ms = (time.perf_counter() - t) / 10 * 1e3
            rehearsed[str(n_ranks)] = dict(hypotheses_of_this_share=int(len(mine)), ms_per_step_of_one_share_without_gather=ms,
                                           evals_per_s_if_every_rank_does_alike=P * T / (ms * 1e-3))
    # algorithmic bytes of a step on this rank: per group of two passes (8 hypotheses) the 2^d*S template rows of every distinct
    # cell in it, per pass its 4 log mu rows written and staged once (8 B each way per bin and hypothesis), one list entry per non-empty bin of every
    # dataset -- once per PASS, not per hypothesis
    mine = deals[steps - 1][rank]
    NS = 2 ** model.d * model.S
    entry_bytes = int(ctx.get_param('tmm_entry_bytes'))
    n_pass = -(-len(mine) // 4)
    from blueice_amd.sharding import cell_ids
    ids = np.sort(cell_ids(model.anchor_z, work[steps - 1][0][mine]))
    cells_in_passes = sum(len(np.unique(ids[i:i + 8])) for i in range(0, len(ids), 8))
    nbytes = 8.0 * NS * model.B * cells_in_passes + 2 * 8.0 * model.B * len(mine) + float(entry_bytes) * nnz * n_pass
    roof = dict(bound='hbm', unit='GB/s', peak=HBM_PEAK_GBS, bytes_per_call=nbytes, list_entry_bytes=entry_bytes, launches_per_call=int(n_launch),
                passes=n_pass, cells_in_pass_groups=int(cells_in_passes), kernel_ms_per_call=kernel_ms,
                achieved=nbytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None,
                achieved_over_whole_step=nbytes / (step_ms * 1e-3) / 1e9, traffic=None,
                kernels='k_morph_logmu_multi<8> (HBM-bound; a cell\'s rows read once per two passes), k_dataset_dot_multi<4,2,2,4,4096> (bound by LDS bank conflicts: 64 lanes x 32 '
                        'random bytes per entry; profiles/r05_toy_points.json), k_dataset_finish_multi<4>',
                note='algorithmic bytes per call / summed kernel time of the call (HIP events); what the call is measured by is '
                     'evaluations per second')
    roof['frac'] = roof['achieved'] / HBM_PEAK_GBS if roof['achieved'] else None
    return dict(workload='C3: 10^4 toy datasets (drawn on the device) x %d hypotheses per step (%d grid cells x %d signal strengths), '
                         'hypotheses dealt over the ranks by grid cell, every rank holds all datasets' % (P, max(1, P // 8), P // max(1, P // 8)),
                scaling='strong', datasets=T, hypotheses=P, steps=steps, value=P * T * steps / elapsed, unit='evals/s', ms_per_step=step_ms,
                results='the [hypotheses][datasets] matrix left in HBM (status words read back); value_results_to_host: the same step with '
                        'the matrix copied to the host as well',
                value_results_to_host=P * T * steps / elapsed_fetch, ms_per_step_results_to_host=elapsed_fetch / steps * 1e3,
                generate_s=gen_s, hypotheses_per_rank_min_max=[min(len(d) for d in deals[0]), max(len(d) for d in deals[0])],
                max_rel_diff_vs_single_point_calls=worst, nonempty_bins=nnz, gather=ranks.kind, roofline=roof,
                evals_per_s_per_gpu_by_hypotheses_per_call=per_gpu,
                four_hypotheses_over_one_per_call=per_gpu['4'] / per_gpu['1'],
                step_split_ms=dict(kernels_busiest_rank=kernel_ms, everything_else=max(0.0, step_ms - kernel_ms)),
                predicted_ceiling=predicted_ceiling(P * T, step_ms, kernel_ms, world), rehearsed_share=rehearsed)


def c5_leg(ctx, ranks, steps=24, threads=8):
    """configs[4] on one grid cell of its anchor grid (2^4 anchors, 6 sources, 50^4 bins, Beeston-Barlow on source 0;
    blueice/likelihood.py:618-660), uploaded into `ctx` in place of the C2 model:
      kernel   `k_morph_reduce<1,true,true>`, one evaluation per launch -- all 113 stream rows, 5.65 GB per pass --
               against the HBM roofline (every rank measures, rank 0 reports);
      scan     256 scan points in that cell, STRONG scaling over the ranks (the host planner groups 8 points per pass)."""
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('C5-2anchor', bb_source=0)
    m.upload(ctx, threads=min(8, threads))
    ctx.set_param('sparse', 0)
    ctx.upload_counts(m.counts(dense=True))
    z, r = m.random_points(4, seed=2)
    plans = [ctx.plan(z[i], r[i]) for i in range(4)]
    for i in range(8):                     # warm-up: the first passes over a freshly uploaded 4.8 GB tensor run slow
        plans[i % 4].run()
    ctx.sync()
    ctx.profile(True)
    for i in range(steps):
        plans[i % 4].run()
    n, ms = ctx.profile_read()
    ctx.profile(False)
    nbytes = plans[0].bytes
    assert nbytes == 8 * (16 * 6 + 16 + 1) * m.B
    gbs = nbytes * n / (ms * 1e-3) / 1e9
    ll, st = plans[0].read()
    for p in plans:
        p.close()
    kern = dict(workload='C5 grid cell: Beeston-Barlow, 6 sources, 2^4 anchors, 50^4 bins, one evaluation per launch',
                kernel='k_morph_reduce<1,true,true> (Beeston-Barlow, nontemporal loads)', bound='hbm',
                bytes_per_launch=nbytes, avg_launch_us=ms / n * 1e3, achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s',
                frac=gbs / HBM_PEAK_GBS, evals_per_s=n / (ms * 1e-3), status_bits=int(st[0]))
    # value AND gradient with respect to the 4 shape and 6 rate parameters in one pass (k_morph_bbgrad: the chain rule through
    # the per-bin Beeston-Barlow root) -- against 1 + 10 value passes for forward differences, which is what a minimiser
    # without it does (blueice/inference.py:111-124 under scipy's default BFGS)
    ctx.eval_grad(z[:1], r[:1])
    t = time.perf_counter()
    for i in range(6):
        gl, gz, gs, gst = ctx.eval_grad(z[i % 4:i % 4 + 1], r[i % 4:i % 4 + 1])
    grad_ms = (time.perf_counter() - t) / 6 * 1e3
    t = time.perf_counter()
    lo_z = np.array([g[0] for g in m.anchor_z]); hi_z = np.array([g[-1] for g in m.anchor_z])
    z8 = np.clip(np.repeat(z[1:2], 8, axis=0) + 1e-3 * np.arange(8)[:, None], lo_z, hi_z)
    gl8, gz8, gs8, _ = ctx.eval_grad(z8, np.repeat(r[1:2], 8, axis=0))
    grad8_ms = (time.perf_counter() - t) * 1e3
    h = 1e-6                                                 # one slope against a central difference of the value kernel
    zp, zm = z[1].copy(), z[1].copy()
    zp[0] += h; zm[0] -= h
    fd = (ctx.eval(zp, r[1])[0][0] - ctx.eval(zm, r[1])[0][0]) / (2 * h)
    assert abs(fd - gz[0][0]) <= 1e-5 * max(1.0, abs(fd)), (fd, gz[0][0])
    assert gl[0] == ctx.eval(z[1], r[1])[0][0]
    kern['gradient'] = dict(kernel='k_morph_bbgrad<16,8> (value + 10 slopes in one pass)', ms_one_point=grad_ms,
                            ms_eight_points_one_cell=grad8_ms, value_pass_ms=ms / n,
                            forward_difference_equivalent_ms=11 * ms / n,
                            slope_vs_central_difference_rel=float(abs(fd - gz[0][0]) / max(1.0, abs(fd))))
    scan = scan_leg(ctx, ranks, m, 256, 2, 'C5 grid cell: 256 Beeston-Barlow scan points (6 sources, 2^4 anchors, 50^4 bins), '
                    'dealt over the ranks; data: ~10 events in EVERY bin (no 16-bin tile without events: the kernel\'s full per-bin '
                    'epilogue everywhere)', sample=1)
    if ranks.world == 1:
        # the same scan on data as sparse as a Beeston-Barlow analysis usually sees (the model's own expectation: ~2 10^4 events in
        # 6.25 10^6 bins): tiles without events take the epilogue's short form (k_scan_bb: the discriminant's zero terms and the
        # logarithm left out, the same bits)
        try:
            counts = m.counts(dense=False)
            ctx.upload_counts(counts)
            zs, rs = m.random_points(256, seed=901)
            ctx.set_param('device_plan_min', 1)
            plan = ctx.plan(zs, rs)
            plan.run(); ctx.sync()
            t = time.perf_counter()
            for _ in range(3):
                plan.run()
            ctx.sync()
            dt = (time.perf_counter() - t) / 3
            st = plan.status()
            plan.close()
            scan['sparse_data'] = dict(events=int(counts.sum()), value=256 / dt, unit='evals/s', ms_per_step=dt * 1e3, status_or=int(st),
                                       tiles_without_events=float((counts[:m.B // 16 * 16].reshape(-1, 16).sum(axis=1) == 0).mean()),
                                       k_scan_bb_launches=int(ctx.get_param('n_bb_scan_launches')))
        except Exception as e:                                  # a side figure: never lose the line over it
            scan['sparse_data'] = {'error': repr(e)}
    return kern, scan


def unbinned_leg(ctx, model, n_events=1000000, steps=24):
    """The extended unbinned likelihood on the same machinery (UnbinnedLogLikelihood, blueice/likelihood.py:528-573,
    :678-690), C2-shaped: 4 sources, 5^3 anchors, 10^6 events.
      set_data   the events are scored at every anchor model ON THE DEVICE (bi_score_events: the resident C2 rows serve
                 as the sources' density histograms, 'piecewise' lookup; only the 3 x N coordinates cross PCIe) --
                 the reference loops `m.score_events(d)` over the 125 anchor models on the host (:557-560);
      evaluate   `k_morph_reduce<1,false,true,2>`: 8 evaluations per launch in grid cells that share no anchor, each
                 streams the 2^3 * 4 rows of pdf values at the events: 8 * 2^d * S * N bytes per evaluation, HBM-bound."""
    from blueice_amd.device import DeviceContext
    rng = np.random.default_rng(77)
    edges = [np.linspace(0.0, 1.0, b + 1) for b in model.bins]
    coords = [rng.uniform(0.0, 1.0, n_events) for _ in model.bins]
    uctx = DeviceContext(ctx.device)
    try:
        ctx.score_events(uctx, 'piecewise', edges, coords)                 # warm (allocations)
        t = time.perf_counter()
        ctx.score_events(uctx, 'piecewise', edges, coords)
        score_s = time.perf_counter() - t
        # per-event rates: the rows are probabilities per bin (~1e-6), the "densities" of this synthetic exercise
        sets = [model.disjoint_cell_points(parity=i, seed=50 + i) for i in range(4)]
        plans = [uctx.plan(zz, rr) for zz, rr in sets]
        PPS = len(sets[0][0])
        NS = 2 ** model.d * model.S
        nbytes = plans[0].bytes
        assert plans[0].launches == 1 and nbytes == PPS * 8 * (NS + 1) * n_events, (nbytes, PPS)
        nbytes = PPS * 8 * NS * n_events                                   # (the plan's figure counts a counts row; this mode reads none)
        for i in range(8):
            plans[i % 4].run()
        uctx.sync()
        uctx.profile(True)
        for i in range(steps):
            plans[i % 4].run()
        n, ms = uctx.profile_read()
        uctx.profile(False)
        ll, st = plans[0].read()
        assert not st.any() and np.all(np.isfinite(ll))
        for p in plans:
            p.close()
        gbs = nbytes * n / (ms * 1e-3) / 1e9
        return dict(workload='unbinned C2 shape: 4 sources, 5^3 anchors, %d events; %d evaluations per launch in disjoint grid cells' % (n_events, PPS),
                    kernel='k_morph_reduce<1,false,true,2> (extended unbinned likelihood, nontemporal loads, in-launch finish)',
                    bound='hbm', events=n_events, bytes_per_launch=nbytes, avg_launch_us=ms / n * 1e3, achieved=gbs,
                    peak=HBM_PEAK_GBS, unit='GB/s', frac=gbs / HBM_PEAK_GBS, evals_per_s=PPS * n / (ms * 1e-3),
                    set_data_on_device_s=score_s,
                    set_data_note='k_score_locate + k_score_rows fill the [125 anchors][4 sources][%d events] tensor (%.1f GB) from the 3 x N '
                                  'coordinates: %.1f M pdf values per second' % (n_events, 8e-9 * model.A * model.S * n_events,
                                                                               model.A * model.S * n_events / score_s / 1e6))
    finally:
        uctx.close()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: this process -- which has not touched the GPU, nor loaded the
    library -- starts N fresh rank processes of this script (blueice_amd.launch sets RANK / LOCAL_RANK / WORLD_SIZE and
    the rendezvous file), which inherit its stdout: rank 0's JSON line IS this command's output.  Exit code = the first
    non-zero rank's."""
    from blueice_amd import launch
    argv = ['--nproc', str(args.gpus)]
    if args.devices:
        argv += ['--devices', args.devices]
    return launch.main(argv + [os.path.abspath(__file__)] + sys.argv[1:])


def dry_run(args):
    """--dry: everything of an N-rank run that needs no GPU -- launch, rendezvous, dealing, gather, assembly, the JSON
    line -- with a stand-in for the evaluation (a checksum of the point).  For the CPU test of the launch path; the
    line says "dry": true and carries no value."""
    from blueice_amd.comm import connect
    from blueice_amd.sharding import deal_points_by_cell, gather_vector
    from blueice_amd.synthetic import SyntheticModel
    from blueice_amd.comm import describe
    world, rank = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0))

    class DryContext:                                 # what connect() asks of a DeviceContext; there is no GPU here
        stream, device = 0, int(os.environ.get('LOCAL_RANK', 0))

        def device_alloc(self, n):
            return type('Buf', (), {'free': lambda self: None})()

        def sync(self):
            pass

    requested = args.backend or 'rccl'
    if requested == 'rccl' and os.environ.get('BLUEICE_AMD_RCCL'):
        top = connect(DryContext(), backend='rccl', rank=rank, world=world, timeout=30.0)     # the init / agreement path, with a stand-in library
    else:
        top = connect(backend='socket', rank=rank, world=world)
    comm = top.boot if top.kind == 'rccl' else top      # (no device memory in a dry run: the vectors travel through the host channel)
    report = describe(top, requested)
    strict_fallback = args.backend == 'rccl' and top.kind != 'rccl' and world > 1
    model = SyntheticModel.named('C2')
    z, r = model.random_points(20000, seed=900)
    deal = deal_points_by_cell(model.anchor_z, z, world)
    mine = deal[rank]
    parts = gather_vector(z[mine].sum(axis=1) + r[mine].sum(axis=1), [len(d) for d in deal], comm)
    out = np.empty(len(z))
    for idx, vals in zip(deal, parts):
        out[idx] = vals
    np.testing.assert_array_equal(out, z.sum(axis=1) + r.sum(axis=1))
    ranks_seen = comm.all_gather(np.array([float(rank), float(os.environ.get('LOCAL_RANK', -1))]))
    comm.barrier()
    if rank == 0:
        print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'evals/s', 'n_gpus': world, 'dry': True,
                          'steps': args.steps, 'warmup': args.warmup,
                          'config': dict({'host_threads_per_rank': host_thread_share(world),
                                          'host_cores_of_this_process': len(os.sched_getaffinity(0))}, **report),
                          'legs': {'C4': {'points': len(z), 'points_per_rank_min_max': [min(len(d) for d in deal), max(len(d) for d in deal)],
                                          'ranks': ranks_seen[:, 0].tolist(), 'devices': ranks_seen[:, 1].tolist(), 'gather': top.kind}}}),
              flush=True)
    top.close()
    return 4 if strict_fallback else 0


# ---------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=40)
    ap.add_argument('--config', default='C2', help='C2 (headline) | C3 | C4 | C4-dense | C5 | C5-2anchor: the leg that '
                                                    'becomes the JSON line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--no-legs', action='store_true')
    ap.add_argument('--backend', default=None, choices=('rccl', 'socket'),
                    help="gather for N > 1: 'rccl' (direct binding) or 'socket' (host; for rehearsals on a box with fewer GPUs "
                         'than ranks).  Not given: rccl, falling back to sockets together if RCCL cannot start (the line says so); '
                         'given as rccl: RCCL or a non-zero exit')
    ap.add_argument('--devices', default=None, help='self-launched N > 1 only: comma-separated GPU index per rank '
                                                    '(default rank r -> GPU r; "0,0" rehearses two ranks on one GPU)')
    ap.add_argument('--dry', action='store_true', help='no GPU: launch, rendezvous, dealing and gather only (tests)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))          # before anything touches the GPU: the ranks are fresh processes
    if args.dry:
        sys.exit(dry_run(args))

    # stdout must carry exactly one JSON line: native libraries (RCCL's banner) write to fd 1 too, so fd 1 points at
    # stderr until the line is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if args.gpus != world:
        log('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks -- running with %d rank(s)' % (args.gpus, world, world))

    from blueice_amd.device import DeviceContext, default_device
    from blueice_amd.synthetic import SyntheticModel

    ctx = DeviceContext(default_device())
    info = ctx.info()
    threads = host_thread_share(world)              # per rank: upload threads and the library's planner threads
    ctx.set_param('host_threads', threads)
    ranks = Ranks(ctx, args.backend)
    gather_report = ranks.report()                  # (collective) RCCL's own rank count, every rank's GPU, the fallback reason if any

    def emit(result):
        if rank == 0:
            result.setdefault('config', {}).update(gather_report)
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            print(json.dumps(result), flush=True)
            os.dup2(2, 1)

    def leave():
        """--backend rccl was asked for and the socket fallback was taken: the numbers stand (the line says which gather ran),
        the command fails -- an 8-GPU run that silently gathered through the host must not pass for an RCCL run."""
        if ranks.fallback_taken():
            log('rank %d: --backend rccl was requested but the gather fell back to sockets (%s): exit code 4' % (
                rank, gather_report.get('gather_fallback_reason')))
        code = 4 if ranks.fallback_taken() else 0
        ranks.close()
        ctx.close()
        if code:
            sys.exit(code)

    K, W = args.steps, args.warmup

    # ---- the other configurations as the JSON line (manual runs) ------------------------------------------
    if args.config != 'C2':
        if args.config in ('C3', 'C4', 'C4-dense'):
            model = SyntheticModel.named('C2')
            model.upload(ctx, threads=min(4, threads))
            steps = max(1, min(K, 5 if args.config != 'C4-dense' else 2))
            if args.config == 'C3':
                leg = toy_points_leg(ctx, ranks, model, 10000, 32, max(1, min(K, 20)))
            else:
                ctx.set_param('sparse', 1 if args.config == 'C4' else 0)
                ctx.upload_counts(model.counts())
                leg = scan_leg(ctx, ranks, model, 10 ** 6, steps,
                               'C4: 10^6 scan points over the C2 model, dealt by grid cell (%s)' % (
                                   'default path: exact non-empty-bin form' if args.config == 'C4' else 'every bin visited'))
            metric = 'likelihood evals/sec, 4-src 5^3-anchor 100^3-bin model, ' + args.config
        else:
            model = SyntheticModel.named(args.config, bb_source=0)
            t = time.perf_counter()
            model.upload(ctx, threads=min(12, threads))
            log('rank %d: %s resident after %.0f s' % (rank, args.config, time.perf_counter() - t))
            ctx.set_param('sparse', 0)
            ctx.upload_counts(model.counts(dense=True))
            leg = scan_leg(ctx, ranks, model, 256, max(1, min(K, 5)),
                           '%s: Beeston-Barlow, 6 sources, %s anchors, 50^4 bins; 256 points dealt by grid cell' % (
                               args.config, 'x'.join(str(n) for n in model.n_anchor)), sample=1)
            metric = 'Beeston-Barlow likelihood evals/sec, 6-src 50^4-bin model, ' + args.config
        # what bounds the leg, over the WHOLE step (planning, kernels, gather): algorithmic work / step time
        step_s = leg['ms_per_step'] * 1e-3
        if args.config in ('C4', 'C4-dense'):
            bins = model.B if args.config == 'C4-dense' else ctx.get_param('nnz_total')
            tf = 2.0 * (2 ** model.d * model.S) * bins * leg['points'] / step_s / 1e12
            roof = dict(bound='mfma', unit='TFLOP/s', peak=FP64_PEAK_TFLOPS, achieved=tf, frac=tf / FP64_PEAK_TFLOPS, traffic=None,
                        note='fp64 FMA work of the morph (2 * 2^d*S flop per visited bin and point; %d bins visited per point) '
                             'over the whole step; the per-bin logarithm on the vector ALU shares the fp64 units and does '
                             'not overlap with the MFMAs (DESIGN.md section 4)' % bins)
        elif args.config == 'C3':
            nbytes = leg['roofline']['bytes_per_call']       # (the leg's own accounting: rows once + one list entry per non-empty bin)
            gbs = nbytes / step_s / 1e9
            roof = dict(bound='hbm', unit='GB/s', peak=HBM_PEAK_GBS, achieved=gbs, frac=gbs / HBM_PEAK_GBS, traffic=None,
                        list_entry_bytes=leg['roofline'].get('list_entry_bytes'),
                        note='algorithmic bytes of a call over the whole step: the 2^d*S template rows once (log mu) + one list entry '
                             '(list_entry_bytes) per non-empty bin of every dataset; three launches')
        else:
            gbs = leg['streamed_bytes_this_rank'] / step_s / 1e9
            roof = dict(bound='hbm', unit='GB/s', peak=HBM_PEAK_GBS, achieved=gbs, frac=gbs / HBM_PEAK_GBS, traffic=None,
                        note='bytes the work items of this rank stream (a work item = up to 16 points of one grid '
                             'cell = one pass over its 2^d * (S + 1) + 1 rows of 50^4 bins) over the whole step; with many points '
                             'per cell the Beeston-Barlow arithmetic per bin and point (root formula, logarithm) on the fp64 '
                             'vector ALU takes over from HBM as the bound')
        result = {'metric': metric, 'value': leg['value'], 'unit': 'evals/s', 'n_gpus': world, 'steps': leg['steps'],
                  'warmup': 1, 'ms_per_step': leg['ms_per_step'], 'higher_is_better': True, 'scaling': 'strong',
                  'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                  'config': {'workload': leg['workload'], 'device': info['arch']}, 'roofline': roof, 'leg': leg,
                  'cpu_baseline': None}
        emit(result)
        leave()
        return

    # ---- headline ----------------------------------------------------------------------------------------
    model = SyntheticModel.named('C2')
    model.upload(ctx, threads=min(4, threads))
    counts = model.counts()
    ctx.set_param('sparse', 0)          # headline = the dense kernel: every evaluation visits every bin
    sets = [model.disjoint_cell_points(parity=i, seed=1000 * rank + i) for i in range(POOL)]
    PPS = len(sets[0][0])                                # points (evaluations) per step
    # one dataset per point of a step: no byte of a step is used twice (algorithmic = compulsory traffic)
    ctx.upload_counts(np.stack([counts] + [model.counts(dataset=i) for i in range(1, PPS)]))
    plans = [ctx.plan(zz, rr, dataset=np.arange(PPS)) for zz, rr in sets]
    z, r = sets[0]
    NS = 2 ** model.d * model.S
    bytes_per_launch = plans[0].bytes
    assert plans[0].launches == 1 and bytes_per_launch == PPS * 8 * (NS + 1) * model.B

    send, _ = ranks.buffers(K * PPS)
    send.from_host(np.full(K * PPS, np.nan))

    def run_steps(n, base=0):
        for i in range(n):
            plans[i % POOL].run(send.ptr + 8 * PPS * ((base + i) % K))

    run_steps(W)
    if ranks.multi:
        ranks.gather(K * PPS)              # warm-up of the collective as well (channels, kernels)
    ranks.barrier()
    t0 = time.perf_counter()
    run_steps(K)
    ctx.sync()
    gathered = ranks.gather(K * PPS) if ranks.multi else None      # the final gather: the only collective
    ranks.barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    # what the timed region computed (every N): the K * PPS results of THIS rank are read back -- all finite, every step
    # that ran the same plan gave the same bits -- and kept for the oracle's verdict on step 0 (cpu_baseline leg, N = 1)
    timed_out = send.to_host(np.float64, K * PPS).reshape(K, PPS)
    assert np.all(np.isfinite(timed_out)), 'the timed steps left non-finite results (or results that were never written)'
    for i in range(min(K, POOL), K):
        assert np.array_equal(timed_out[i], timed_out[i % POOL]), 'step %d repeats the plan of step %d but not its results' % (i, i % POOL)
    outputs = dict(outputs_checked=True, results_read_back=int(timed_out.size),
                   check='all finite; steps that repeat a plan repeat its bits')
    if gathered is not None:
        # every rank evaluated different points: finite everywhere, and this rank's row is what it computed
        assert gathered.shape == (world, K * PPS) and np.all(np.isfinite(gathered)), 'gathered results are not finite'
        np.testing.assert_array_equal(gathered[rank], timed_out.reshape(-1))

    # kernel time of the same steps, HIP events on the context stream around every launch
    ctx.profile(True)
    n_prof = min(K, 256)
    run_steps(n_prof)
    launches, ms = ctx.profile_read()
    ctx.profile(False)
    achieved = bytes_per_launch * launches / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

    # HBM bytes per launch from the PMC counters (rocprofv3 cannot run inside this process): taken from the
    # committed summary of the same command (profiles/), corrected as MI355X_MICROARCH.md prescribes
    traffic = traffic_src = None
    for name in ('r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                pm = json.load(f)
            if traffic is None and pm.get('algorithmic_bytes_per_launch') == bytes_per_launch:
                traffic = pm['traffic_bytes_per_launch']
                traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)' % name
        except Exception:
            pass

    # the device's read-only ceilings, measured live on the same resident tensor
    stream_ceiling = linear_ceiling = copy_ceiling = None
    if rank == 0:
        try:
            stream_ceiling = max(ctx.stream_bandwidth(items=PPS, rows=NS, nontemporal=True, blocks_per_cu=b, reps=4)
                                 for b in (8, 16, 32))
            linear_ceiling = max(ctx.read_bandwidth(nontemporal=True, blocks_per_cu=b, reps=3) for b in (16, 32))
            copy_ceiling = ctx.copy_bandwidth(1 << 31, reps=3)
        except Exception as e:                       # a measurement aid only: never fail the bench line over it
            log('bench.py: bandwidth probe failed: %s' % e)

    result = None
    if rank == 0:
        result = {
            'metric': METRIC,
            'value': world * K * PPS / elapsed, 'unit': 'evals/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'C2: 4 sources, 3 shape params (5^3 anchors), 100^3 bins; step = one batched call '
                                   '(one kernel launch) of %d independent dense evaluations, each against its own '
                                   'dataset, in grid cells that share no anchor (no byte re-used), tensor '
                                   'replicated per GPU' % PPS,
                       'sources': model.S, 'anchors': list(model.n_anchor), 'bins': list(model.bins),
                       'evals_per_step': PPS, 'device': info['arch'], 'gather': ranks.kind,
                       'host_threads_per_rank': threads},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'k_morph_reduce<1,false,true> (G=1, no BB, nontemporal loads, in-launch finish)',
                         'bytes_per_launch': bytes_per_launch, 'avg_launch_us': ms / max(launches, 1) * 1e3,
                         'step_minus_kernel_us': elapsed / K * 1e6 - ms / max(launches, 1) * 1e3,
                         'stream_ceiling': stream_ceiling, 'linear_read_ceiling': linear_ceiling,
                         'copy_ceiling': copy_ceiling,
                         'ceiling_note': 'stream_ceiling: GB/s of the same launch shape (%d items x %d concurrent rows, 16 or '
                                         '32 B per lane per row, nontemporal, best of three grid sizes) with no arithmetic and no counts; '
                                         'linear_read_ceiling: one linear 16-byte-load sum over the 4 GB tensor; '
                                         'copy_ceiling: bytes read + written per second of a 2 GiB device-to-device '
                                         'hipMemcpy; all in this process' % (PPS, NS)},
        }

    # ---- strong-scaling legs: the configurations that really shard (every N) ---------------------------------
    legs = {}
    if not args.no_legs:
        ctx.set_param('sparse', 1)
        ctx.upload_counts(counts)
        legs['C4'] = scan_leg(ctx, ranks, model, 10 ** 6, 3,
                              'C4: 10^6 scan points over the C2 model, dealt by grid cell (default path: exact '
                              'non-empty-bin form, %d bins with data)' % ctx.get_param('nnz_total'))
        nnz = ctx.get_param('nnz_total')
        per_rank_flops = 2.0 * NS * nnz * legs['C4']['points_per_rank_min_max'][1]
        legs['C4']['roofline_scan'] = {
            'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': FP64_PEAK_TFLOPS,
            'achieved': per_rank_flops / (legs['C4']['ms_per_step'] * 1e-3) / 1e12,
            'note': 'fp64 FMA work of the morph over the %d bins with data (2 * 2^d*S flop per bin and evaluation) of the busiest '
                    'rank over the whole step (device planning + k_scan_mfma on the compacted, count-ordered rows + finish + '
                    'read-back); the logarithms run on the vector ALU, which shares the fp64 units with the matrix cores' % nnz}
        legs['C4']['roofline_scan']['frac'] = legs['C4']['roofline_scan']['achieved'] / FP64_PEAK_TFLOPS
        ctx.set_param('sparse', 0)
        ctx.upload_counts(counts)
        before = ctx.get_param('n_valid_launches')
        legs['C4-dense'] = scan_leg(ctx, ranks, model, 10 ** 6, 1,
                                    'C4: 10^6 scan points over the C2 model, dealt by grid cell (every bin visited: '
                                    'non-empty-bin pass + validity pass of all bins on the fp64 matrix cores, k_scan_valid)')
        legs['C4-dense']['matrix_core_launches'] = ctx.get_param('n_valid_launches') - before
        per_rank_flops = 2.0 * NS * model.B * legs['C4-dense']['points_per_rank_min_max'][1]
        legs['C4-dense']['roofline_scan'] = {
            'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': FP64_PEAK_TFLOPS,
            'achieved': per_rank_flops / (legs['C4-dense']['ms_per_step'] * 1e-3) / 1e12,
            'note': 'fp64 FMA work of the morph alone (2 * 2^d*S * bins flop per evaluation) of the busiest rank over the '
                    'whole step (planning + non-empty-bin pass + k_scan_valid + gather)'}
        legs['C4-dense']['roofline_scan']['frac'] = legs['C4-dense']['roofline_scan']['achieved'] / FP64_PEAK_TFLOPS
        legs['C3'] = toy_points_leg(ctx, ranks, model, 10000, 32, 20)
        legs['C3-one-point'] = toy_leg(ctx, ranks, model, 10000, 100)      # (100 calls of ~0.13 ms: a steadier mean than 20)
        ctx.set_param('sparse', 0)
        ctx.upload_counts(counts)
    if rank == 0:
        result['legs'] = legs

    if rank == 0 and world == 1 and not args.no_extras:
        result['extras'] = extras(ctx, model, counts, z, r, PPS, bytes_per_launch // PPS)
        ex = result['extras']
        # configs[1] read literally -- ONE evaluation of a single dataset per call: the single-point kernel (k_morph_single, the
        # launch behind lf(**kw)) against the same roofline, next to the 8-evaluation launch the headline is quoted on, and
        # against the read ceiling of its own access pattern (one item x 32 concurrent rows, nontemporal, best of 8 passes)
        one_us = ex['sync_call_split_us']['kernel_by_hip_events']
        one_gbs = (bytes_per_launch // PPS) / (one_us * 1e-6) / 1e9
        one_ceiling = max(ctx.stream_bandwidth(items=1, rows=NS, nontemporal=True, blocks_per_cu=b, reps=8) for b in (4, 8))
        result['roofline']['single_evaluation'] = {
            'kernel': 'k_morph_single<false,true,0,true> (one launch from templates to scalar, in-launch finish)',
            'bytes_per_launch': bytes_per_launch // PPS, 'avg_launch_us': one_us, 'achieved': one_gbs, 'frac': one_gbs / HBM_PEAK_GBS,
            'stream_ceiling_one_item': one_ceiling, 'frac_of_its_stream_ceiling': one_gbs / one_ceiling,
            'wall_us_per_call': ex['sync_call_latency_us'],
            'note': 'a single 264 MB pass lasts ~44 us: wave dispatch, the first loads\' latency and the in-launch finish do not '
                    'amortise as they do over the 8-evaluation launch; the bare read of the same rows in the same order reaches '
                    'stream_ceiling_one_item'}
        result['north_star'] = {
            'hbm_frac_target': 0.70, 'hbm_frac': result['roofline']['frac'],
            'evals_per_s_target': 1e6,
            'dense_evals_per_s_ceiling_no_reuse': HBM_PEAK_GBS * 1e9 / (bytes_per_launch // PPS),
            'evals_per_s_scan_every_bin_visited': ex.get('dense_scan_131072_evals_per_s'),
            'evals_per_s_scan_default_path_incl_planning_and_gather': legs.get('C4', {}).get('value'),
            'evals_per_s_toy_mc_10000': legs.get('C3', {}).get('value'),
            'note': 'an evaluation that shares no template bytes with its neighbours moves %.0f MB, so 8 TB/s caps it at '
                    '%.1f k/s: `value` is that case, at `roofline.frac` of the peak.  10^6/s needs re-use, and is met '
                    'where the path re-uses: a 10^6-point scan on the default path (exact non-empty-bin identity, '
                    'templates >= 0) and 10^4 toys per call -- both measured end to end under `legs`.  With EVERY bin '
                    'visited a scan is bound by the fp64 units (morph FMAs on the matrix cores + per-bin terms), see '
                    'legs.C4-dense.roofline_scan' % (bytes_per_launch / PPS / 1e6,
                                                     HBM_PEAK_GBS * 1e9 / (bytes_per_launch / PPS) / 1e3),
        }

    if rank == 0:
        result.update(outputs)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(model, counts, (z, r))
        # the oracle's value of evaluation 0 of step 0 (point z[0], dataset 0) against what the timed region wrote there
        want = result['cpu_baseline'].pop('oracle_value')
        rel = abs(timed_out[0, 0] - want) / max(1.0, abs(want))
        assert rel <= 1e-10, 'timed step 0, evaluation 0: %r on the device, %r by the oracle' % (timed_out[0, 0], want)
        result['outputs_rel_diff_vs_oracle_step0_eval0'] = rel
        result['check'] = outputs['check'] + '; evaluation 0 of timed step 0 equals the oracle to 1e-10'
        result['cpu_baseline']['host_cores_available'] = os.cpu_count()
        calls = result.get('extras', {}).get('api_bestfit_scipy_likelihood_calls')
        if calls:                # the same fit on the host = that many evaluations at the measured single-thread rate
            result['cpu_baseline']['bestfit_scipy_estimate_s'] = calls / result['cpu_baseline']['value']
            result['cpu_baseline']['bestfit_scipy_estimate_note'] = (
                '%d likelihood calls of the device fit (extras.api_bestfit_scipy_s) x the single-thread time per '
                'evaluation measured above; an estimate, the fit itself is not run on the host' % calls)
        try:
            share = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
            n_procs = max(1, min(16, share))
            result['cpu_baseline']['all_cores'] = cpu_baseline_all_cores('C2', n_procs)
            result['cpu_baseline']['all_cores']['cores_note'] = (
                'a 1-GPU box of this pool is granted a CPU share of 16 workers (the pool kills runs with larger worker '
                'pools), although the host reports %d logical cores; rate scales ~linearly with processes' % (os.cpu_count() or 0))
        except Exception as e:                                       # never let the side figure break the line
            result['cpu_baseline']['all_cores'] = {'error': repr(e)}
    elif rank == 0:
        result['cpu_baseline'] = None

    if rank == 0 and world == 1 and not args.no_legs:
        try:
            result['legs']['unbinned'] = unbinned_leg(ctx, model)
            result['roofline_unbinned'] = {k: result['legs']['unbinned'][k] for k in
                                           ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'bytes_per_launch', 'avg_launch_us')}
        except Exception as e:                                  # a side leg: never lose the line over it
            result['legs']['unbinned'] = {'error': repr(e)}

    # ---- configs[4], Beeston-Barlow: kernel roofline + strong-scaling scan on one grid cell, every N.  Last: the C5
    # cell's tensors replace the C2 model in this rank's context (same stream, same communicator, same gather buffers)
    for p in plans:
        p.close()
    if not args.no_legs:
        kern, scan = c5_leg(ctx, ranks, threads=threads)
        if rank == 0:
            result['legs']['C5-BB'] = kern
            result['legs']['C5-BB-scan'] = scan
            result['roofline_bb'] = {k: kern[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'bytes_per_launch', 'avg_launch_us')}

    emit(result)
    leave()


# ---------------------------------------------------------------------------------------------------------
def extras(ctx, model, counts, z, r, PPS, bytes_per_eval):
    """Other call shapes of the same path, N = 1 (state on entry: sparse = 0, `counts` resident as dataset 0)."""
    ex = {}
    ctx.set_param('sparse', 0)
    ctx.upload_counts(counts)

    def timed(fn, reps):
        fn()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t) / reps

    zz, rr = model.stratified_points(seed=3)           # all 64 cells in one call: neighbours share corners
    p = ctx.plan(zz, rr)
    ex['all_64_cells_batch_evals_per_s'] = len(zz) / timed(p.run, 20)
    p.close()
    singles = [ctx.plan(z[i], r[i]) for i in range(PPS)]     # one evaluation per launch, rotating cells
    ex['one_point_per_launch_evals_per_s'] = PPS / timed(lambda: [q.run() for q in singles], 8)
    for q in singles:
        q.close()
    p = ctx.plan(z[0], r[0])                           # same cell every call (a fit's access pattern)
    dt = timed(p.run, 1000)
    ex['same_cell_evals_per_s'] = 1 / dt
    ex['same_cell_GBps'] = bytes_per_eval / dt / 1e9
    p.close()

    # the synchronous call `lf(**kw)` makes inside a minimizer: bi_eval(P = 1), rotating cells, and where its time goes
    for i in range(20):
        ctx.eval_one(z[i % PPS], r[i % PPS])
    ctx.set_param('single_timing_reset', 1)
    t = time.perf_counter()
    for i in range(400):
        ctx.eval_one(z[i % PPS], r[i % PPS])
    wall = (time.perf_counter() - t) / 400 * 1e6
    n = max(ctx.get_param('single_calls'), 1)
    split = {k: ctx.get_param('single_ns_' + k) / n / 1e3 for k in ('host', 'launch', 'wait')}
    ctx.profile(True)
    for i in range(64):
        ctx.eval_one(z[i % PPS], r[i % PPS])
    nl, kms = ctx.profile_read()
    ctx.profile(False)
    ex['sync_call_latency_us'] = wall
    ex['sync_call_split_us'] = {'python_ctypes': wall - sum(split.values()), 'host_geometry': split['host'],
                                'launch_calls': split['launch'], 'wait_for_result': split['wait'],
                                'kernel_by_hip_events': kms / max(nl, 1) * 1e3,
                                'note': 'wait_for_result = launch latency + kernel + in-launch finish + result word reaching the host'}
    for i in range(20):
        ctx.eval_one(z[0], r[0] * (1 + 1e-3 * i))
    t = time.perf_counter()
    for i in range(400):
        ctx.eval_one(z[0], r[0] * (1 + 1e-4 * i))
    ex['sync_call_same_cell_latency_us'] = (time.perf_counter() - t) / 400 * 1e6

    zz, rr = model.random_points(16384, seed=7)        # scan batch: cell-grouped, templates reused
    p = ctx.plan(zz, rr)
    ex['scan_batch_16384_evals_per_s'] = 16384 / timed(p.run, 3)
    p.close()
    zz, rr = model.random_points(131072, seed=11)      # the same on a scan of 131 072 points (128 items per cell)
    p = ctx.plan(zz, rr)
    dt = timed(p.run, 2)
    ex['dense_scan_131072_evals_per_s'] = len(zz) / dt
    ex['dense_scan_131072_fp64_fma_TFLOPs'] = 2.0 * 32 * model.B * len(zz) / dt / 1e12
    p.close()
    dense_counts = model.counts(dense=True)            # ~10 events per bin: a logarithm in every bin
    ctx.upload_counts(dense_counts)
    p = ctx.plan(zz, rr)
    ex['dense_scan_131072_dense_data_evals_per_s'] = len(zz) / timed(p.run, 1)
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
    t = time.perf_counter()
    p = ctx.plan(zz, rr)
    t_plan = time.perf_counter() - t
    dt = timed(p.run, 3)
    ex['sparse_scan_131072_evals_per_s_device'] = len(zz) / dt
    ex['sparse_scan_131072_evals_per_s_incl_planning'] = len(zz) / (dt + t_plan)
    ex['sparse_nonempty_bins'] = ctx.get_param('nnz_total')
    p.close()
    t = time.perf_counter()
    for i in range(400):
        ctx.eval_one(z[i % PPS], r[i % PPS])
    ex['sync_call_latency_sparse_form_us'] = (time.perf_counter() - t) / 400 * 1e6
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
    lf.ctx.set_param('single_timing_reset', 1)
    t = time.perf_counter()
    best, ll = lf.bestfit_scipy(**fixed)
    ex['api_bestfit_scipy_s'] = time.perf_counter() - t
    ex['api_bestfit_scipy_likelihood_calls'] = int(lf.ctx.get_param('single_calls'))
    t = time.perf_counter()
    lf.bestfit_scipy(use_gradient=True, **fixed)
    ex['api_bestfit_scipy_with_gradient_s'] = time.perf_counter() - t
    lf.bestfit_scipy(batch_stencil=False, **fixed)
    t = time.perf_counter()
    lf.bestfit_scipy(batch_stencil=False, **fixed)          # the reference's stream of scalar calls (round 2's number)
    ex['api_bestfit_scipy_scalar_stream_s'] = time.perf_counter() - t
    lf.bestfit_batched(**fixed)
    t = time.perf_counter()
    eb, ell = lf.bestfit_batched(**fixed)                  # the same fit as ONE problem of the batched engine (kinks handled, 5 starts)
    ex['api_bestfit_batched_single_fit_s'] = time.perf_counter() - t
    ex['api_bestfit_batched_single_fit_max_loglikelihood'] = float(ell[0])
    assert ell[0] >= ll - 1e-6 * abs(ll), (ell[0], ll)
    t = time.perf_counter()
    for i in range(300):
        lf(shape0=0.1 + 1e-4 * i, s0_rate_multiplier=1.05)
    ex['api_call_us'] = (time.perf_counter() - t) / 300 * 1e6
    # profiled scan (blueice/inference.py:392-443 with floating nuisances): 1024 hypotheses of shape0, at each of them the
    # first rate and the two other shape parameters fitted -- all 1024 fits advance together on the batched engine
    grid = np.linspace(-1.9, 1.9, 1024)
    lf.bestfit_batched(points={'shape0': grid[:64]}, **fixed)
    t = time.perf_counter()
    _, prof_ll, info = lf.bestfit_batched(points={'shape0': grid}, return_info=True, **fixed)
    dt = time.perf_counter() - t
    ex['api_profiled_scan_1024_points_s'] = dt
    ex['api_profiled_points_per_s'] = len(grid) / dt
    ex['api_profiled_scan_device_calls'] = int(info['calls'])
    ex['api_profiled_scan_evaluations'] = int(info['evaluations'])
    ex['api_profiled_scan_converged_fraction'] = float(np.mean(info['converged'] | info['stalled']))
    k = int(np.argmax(prof_ll))
    t = time.perf_counter()
    _, one = lf.bestfit_scipy(use_gradient=True, shape0=float(grid[k]), **fixed)
    ex['api_profiled_point_sequential_fit_s'] = time.perf_counter() - t       # what ONE of those fits costs on its own
    # (never below the sequential fit; above it where scipy's single start stops at a kink of the morph -- the base
    #  values of the shape parameters ARE anchors -- which the engine steps off on the side that goes down)
    assert prof_ll[k] >= one - 1e-6 * abs(one), (one, prof_ll[k])
    ex['api_profiled_point_engine_minus_sequential_ll'] = float(prof_ll[k] - one)
    t = time.perf_counter()
    up = lf.one_parameter_interval('s0_rate_multiplier', bound=3.0, kind='upper', confidence_level=0.9, **fixed)
    ex['api_upper_limit_s'] = time.perf_counter() - t
    ex['api_upper_limit_value'] = up
    ex['api_bestfit_max_loglikelihood'] = ll
    # toy-MC with a fit per toy -- the reference: `d = lf.base_model.simulate(); lf.set_data(d); bestfit_scipy(lf)`, one toy
    # after the other (blueice/model.py:69-91, inference.py:131-178).  Here 256 toys are drawn on the device and all of
    # them fitted at the same time on the batched engine (one problem per dataset, bi_eval_grad with a dataset per point)
    n_toys = 256
    t = time.perf_counter()
    lf.simulate_toys(n_toys, seed=99)
    ex['api_toy_fits_generate_s'] = time.perf_counter() - t
    lf.bestfit_toys(0, 32, **fixed)
    t = time.perf_counter()
    _, toy_ll, tinfo = lf.bestfit_toys(return_info=True, **fixed)
    dt = time.perf_counter() - t
    ex['api_toy_fits_256_s'] = dt
    ex['api_toy_fits_per_s'] = n_toys / dt
    ex['api_toy_fits_device_calls'] = int(tinfo['calls'])
    ex['api_toy_fits_evaluations'] = int(tinfo['evaluations'])
    toy7 = lf.ctx.download_counts(7)
    t = time.perf_counter()
    lf.set_binned_data(toy7.reshape(model.bins))
    _, one = lf.bestfit_scipy(**fixed)
    ex['api_toy_fit_sequential_s'] = time.perf_counter() - t                  # set_data + fit of ONE toy, the loop body
    assert toy_ll[7] >= one - 1e-6 * abs(one), (toy_ll[7], one)
    # ... and an ensemble of 2048 toys drawn and fitted in chunks of 256 (the toys' compacted templates: 10.5 GB per chunk)
    t = time.perf_counter()
    ens, ens_ll = lf.toy_mc_fits(2048, chunk=256, seed=100, **fixed)
    dt = time.perf_counter() - t
    ex['api_toy_mc_2048_draw_and_fit_s'] = dt
    ex['api_toy_mc_fits_per_s'] = 2048 / dt
    ex['api_toy_mc_fitted_rate_mean_std'] = [float(np.mean(ens['s0_rate_multiplier'])), float(np.std(ens['s0_rate_multiplier']))]
    assert np.all(np.isfinite(ens_ll))
    lf.set_binned_data(counts.reshape(model.bins))
    # template building: the binning of one source's Monte Carlo sample (10^6 events, 3 dimensions, 100^3 bins) -- what
    # prepare() does once per source and anchor model (blueice/source.py:287-299)
    rng = np.random.default_rng(21)
    edges = [np.linspace(-4, 4, 101)] * 3
    sample = [rng.normal(size=1000000) for _ in range(3)]
    ctx.histogram_events(edges, sample)
    t = time.perf_counter()
    on_device = ctx.histogram_events(edges, sample)
    ex['template_histogram_1e6_events_device_ms'] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    on_host = np.histogramdd(np.stack(sample, 1), bins=edges)[0]
    ex['template_histogram_1e6_events_numpy_ms'] = (time.perf_counter() - t) * 1e3
    assert np.array_equal(on_device, on_host)
    zz, rr = model.random_points(1000000, seed=11)       # configs[3] through lf.eval_points: dict of arrays in, ll [P] out
    pts = {'shape%d' % i: zz[:, i] for i in range(zz.shape[1])}
    pts.update({'s%d_rate_multiplier' % s: rr[:, s] for s in range(model.S)})
    lf.eval_points(pts)
    t = time.perf_counter()
    out = lf.eval_points(pts)
    ex['api_eval_points_1e6_s'] = time.perf_counter() - t
    assert out.shape == (1000000,) and np.all(np.isfinite(out))
    lf.ctx.set_param('sparse', 0)                       # the same fit with every bin visited on every call
    lf.set_binned_data(counts.reshape(model.bins))
    lf.bestfit_scipy(**fixed)
    t = time.perf_counter()
    _, ll_dense = lf.bestfit_scipy(**fixed)
    ex['api_bestfit_scipy_every_bin_visited_s'] = time.perf_counter() - t
    ex['api_bestfit_every_bin_visited_max_loglikelihood'] = ll_dense
    del lf
    return ex


if __name__ == '__main__':
    main()
