"""Start one process per GPU:  python -m blueice_amd.launch --nproc N [--port P] script.py [args ...]

Every rank is a fresh `python script.py args` child with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
and BLUEICE_AMD_RDZV (the rendezvous file of blueice_amd.comm) in its environment, started before anything has
touched the GPU (this launcher never does).  Exit code = the first non-zero exit code of a rank; when one rank
fails the others are terminated.  A launcher that sets the same variables (torch.distributed.run) works as well:
blueice_amd.comm only reads the environment.
"""
import argparse
import os
import signal
import subprocess
import sys
import tempfile
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog='python -m blueice_amd.launch')
    ap.add_argument('--nproc', type=int, required=True)
    ap.add_argument('--port', type=int, default=29511)
    ap.add_argument('--devices', default=None, help='comma-separated GPU index per rank (default: rank r -> GPU r)')
    ap.add_argument('script')
    ap.add_argument('args', nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    devices = [int(x) for x in a.devices.split(',')] if a.devices else list(range(a.nproc))
    if len(devices) != a.nproc:
        ap.error('--devices needs one entry per rank')
    fd, rdzv = tempfile.mkstemp(prefix='blueice_amd_rdzv_')
    os.close(fd)
    os.unlink(rdzv)
    procs = []
    # the ranks import what the launcher can import: its working directory (as `python -m` has it on the path) and the
    # directory this package lives in come first on their PYTHONPATH
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.pathsep.join([p for p in (os.getcwd(), here, os.environ.get('PYTHONPATH', '')) if p])
    for r in range(a.nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(devices[r]), WORLD_SIZE=str(a.nproc),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(a.port), BLUEICE_AMD_RDZV=rdzv, PYTHONPATH=path)
        procs.append(subprocess.Popen([sys.executable, a.script] + a.args, env=env))
    code = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and code == 0:
                    code = rc
                    for q in live:                      # a rank failed: the others would wait for it forever
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        code = 130
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
    finally:
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        try:
            os.unlink(rdzv)
        except OSError:
            pass
    return code


if __name__ == '__main__':
    sys.exit(main())
