"""`python bench.py --gpus N` launches its own ranks ("Next round" 1): the parent -- before anything
touches the GPU -- starts N fresh processes of the script, rank 0's JSON line is the command's output, and a failing
rank fails the command.  Run here without a GPU in the bench's --dry mode (launch, rendezvous, dealing, gather, assembly,
JSON; a checksum stands in for the evaluation), and -- marked gpu -- for real with two ranks sharing the box's one GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'BLUEICE_AMD_RDZV')}
    env.update(extra)
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bare_invocation_launches_n_ranks():
    res = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry', '--steps', '2', '--warmup', '1'], env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    assert line['n_gpus'] == 2 and line['dry'] is True and line['value'] is None
    leg = line['legs']['C4']
    assert leg['ranks'] == [0.0, 1.0] and leg['devices'] == [0.0, 1.0]          # rank r -> GPU r
    assert sum(leg['points_per_rank_min_max']) == leg['points']                  # both ranks' shares arrived


def test_devices_option_and_launcher_environment():
    res = subprocess.run([sys.executable, BENCH, '--gpus', '3', '--dry', '--devices', '0,0,0'], env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert _json_line(res.stdout)['legs']['C4']['devices'] == [0.0, 0.0, 0.0]
    # under a launcher (the environment is already there) the script does NOT start ranks of its own
    res = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--dry'], env=_clean_env(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and _json_line(res.stdout)['n_gpus'] == 1


def test_host_threads_are_shared_among_the_ranks():
    """Eight ranks on a node must not each start a full set of upload / planner threads: a rank's share is the cores this
    process may run on divided by the number of ranks ("Next round" 3)."""
    cores = len(os.sched_getaffinity(0))
    res = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--dry'], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    assert line['n_gpus'] == 8 and len(line['legs']['C4']['ranks']) == 8
    assert line['config']['host_threads_per_rank'] == max(1, min(16, cores // 8))
    assert line['config']['host_threads_per_rank'] * 8 <= max(8, cores)
    sys.path.insert(0, ROOT)
    import bench
    assert bench.host_thread_share(1) == min(16, cores) and bench.host_thread_share(10 ** 6) == 1
    # the strong-scaling legs print what N ranks can reach from one run's own split of a step
    pc = bench.predicted_ceiling(10 ** 6, step_ms=12.0, kernel_ms=9.0, world=1)
    assert pc['fixed_ms_per_rank'] == 3.0 and abs(pc['evals_per_s']['8'] - 1e6 / ((3.0 + 9.0 / 8) * 1e-3)) < 1e-3
    pc2 = bench.predicted_ceiling(10 ** 6, step_ms=3.0 + 9.0 / 4, kernel_ms=9.0 / 4, world=4)      # the same leg measured at N = 4
    assert abs(pc2['evals_per_s']['8'] - pc['evals_per_s']['8']) < 1e-3 and abs(pc2['evals_per_s']['1'] - pc['evals_per_s']['1']) < 1e-3
    # the toy leg: the log mu pass over all bins is repeated by every rank in full, only the rest of the kernels divides
    pc3 = bench.predicted_ceiling(10 ** 4, step_ms=0.14, kernel_ms=0.10, world=1, undivided_kernel_ms=0.047)
    assert abs(pc3['fixed_ms_per_rank'] - 0.087) < 1e-12 and abs(pc3['evals_per_s']['8'] - 1e4 / ((0.087 + 0.053 / 8) * 1e-3)) < 1e-3
    assert abs(pc3['evals_per_s']['1'] - 1e4 / 0.14e-3) < 1e-3


def _fake_rccl(tmp_path):
    """tests/native/fake_rccl.c as a shared library: the entry points blueice_amd/comm.py binds, no GPU behind them."""
    so = str(tmp_path / 'libfake_rccl.so')
    res = subprocess.run(['gcc', '-O1', '-shared', '-fPIC', '-Wall', '-Werror', '-o', so, os.path.join(ROOT, 'tests', 'native', 'fake_rccl.c')],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return so


def test_the_line_says_what_rccl_saw(tmp_path):
    """The first 8-GPU run must verify itself ("Next round" 6): the line carries the number of ranks RCCL
    itself reports (ncclCommCount), every rank's GPU as RCCL sees it (ncclCommCuDevice), the gather that ran and the reason for
    a fallback; and `--backend rccl` given explicitly ends non-zero when the socket fallback had to be taken."""
    so = _fake_rccl(tmp_path)
    res = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--dry', '--backend', 'rccl'], env=_clean_env(BLUEICE_AMD_RCCL=so),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    cfg = _json_line(res.stdout)['config']
    assert cfg['backend_requested'] == 'rccl' and cfg['gather_kind'] == 'rccl' and cfg['rccl_ranks'] == 4
    assert cfg['rank_devices'] == [0, 1, 2, 3] and cfg['rank_devices_source'] == 'ncclCommCuDevice'
    assert cfg['gather_fallback_reason'] is None and cfg['rccl_version'] == 99999
    # ncclCommInitRank fails on one rank: every rank falls back together, the line says why -- and the command fails,
    # because RCCL was asked for by name
    res = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry', '--backend', 'rccl'],
                         env=_clean_env(BLUEICE_AMD_RCCL=so, FAKE_RCCL_INIT='fail', FAKE_RCCL_BAD_RANK='1'), capture_output=True, text=True, timeout=300)
    assert res.returncode == 4, (res.returncode, res.stderr[-2000:])
    cfg = _json_line(res.stdout)['config']
    assert cfg['gather_kind'] == 'socket' and cfg['rccl_ranks'] is None and 'ncclCommInitRank failed' in cfg['gather_fallback_reason']
    assert cfg['rank_devices'] == [0, 1] and cfg['rank_devices_source'] == 'launcher'
    # the same failure without --backend: the agreed fallback is allowed, the line still says it happened
    res = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry'],
                         env=_clean_env(BLUEICE_AMD_RCCL=so, FAKE_RCCL_INIT='fail', FAKE_RCCL_BAD_RANK='1'), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    cfg = _json_line(res.stdout)['config']
    assert cfg['gather_kind'] == 'socket' and 'ncclCommInitRank failed' in cfg['gather_fallback_reason']


def test_a_failing_rank_fails_the_command():
    """Without a GPU every rank dies in DeviceContext(): the command must end non-zero and say why -- never print a line."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip('a GPU is present')
    except ImportError:
        pass
    res = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-legs', '--no-extras',
                          '--no-cpu-baseline', '--backend', 'socket'], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert 'no CPU fallback' in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith('{')]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_end_to_end():
    """The real thing on the GPU box: `bench.py --gpus 2` with both ranks on GPU 0 and the host gather (RCCL refuses two
    ranks on one device) -- headline plus every strong-scaling leg, with the scans dealt on the device."""
    res = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--devices', '0,0', '--backend', 'socket', '--steps', '10',
                          '--warmup', '2'], env=_clean_env(), capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-4000:]
    line = _json_line(res.stdout)
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['scaling'] == 'weak'
    legs = line['legs']
    assert set(legs) >= {'C4', 'C4-dense', 'C3', 'C5-BB', 'C5-BB-scan'}
    assert legs['C4']['dealing'].startswith('device planner') and legs['C4']['points'] == 10 ** 6
    assert sum(legs['C4']['points_per_rank_min_max']) == 10 ** 6
    assert legs['C5-BB-scan']['dealing'].startswith('device planner') and sum(legs['C5-BB-scan']['points_per_rank_min_max']) == 256
    assert legs['C3']['datasets'] == 10000
    # (value: the gathered matrix left in HBM -- through the HOST gather of this two-ranks-on-one-GPU run that is a copy down and
    #  up again, so it need not beat the fetched form here as it does with one rank or with RCCL)
    assert legs['C3']['value_results_to_host'] > 0 and legs['C3']['value'] > 0
    for leg in ('C4', 'C4-dense', 'C5-BB-scan'):
        assert legs[leg]['sample_max_rel_diff_vs_single_point_kernel'] <= 1e-11
    assert line['outputs_checked'] is True
    assert line['config']['host_threads_per_rank'] == max(1, min(16, len(os.sched_getaffinity(0)) // 2))
    for leg in ('C4', 'C4-dense', 'C3', 'C5-BB-scan'):
        pc = legs[leg]['predicted_ceiling']
        assert pc['measured_at_n'] == 2 and pc['evals_per_s']['8'] >= pc['evals_per_s']['1'] > 0
    assert legs['C3']['roofline']['bound'] == 'hbm' and legs['C3']['roofline']['launches_per_call'] >= 1
