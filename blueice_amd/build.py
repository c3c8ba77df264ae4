"""Build libblueice_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m blueice_amd.build [--force]

build_host() compiles csrc/host_backend.cpp -- the minimal entry points of the same C ABI as plain C++ loops on the
host -- into lib/libblueice_host.so.  That library is a BOUNDARY TEST BUILD: the package
never loads it (blueice_amd/_capi.py binds libblueice_hip.so only); tests and tools/ load it by explicit path.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SRC = os.path.join(CSRC, 'blueice_hip.hip')          # the one translation unit; bi_*.h are included by it
HDR = os.path.join(os.path.dirname(_HERE), 'include', 'blueice_hip.h')
OUT_DIR = os.path.join(_HERE, 'lib')
OUT = os.path.join(OUT_DIR, 'libblueice_hip.so')
ARCH = 'gfx950'


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return 'hipcc'


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT):
        deps = [HDR] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))]
        newest = max(os.path.getmtime(f) for f in deps)
        if os.path.getmtime(OUT) >= newest:
            return OUT
    cmd = [hipcc(), '-O3', '-std=c++17', '--offload-arch=' + ARCH, '-shared', '-fPIC',
           # no implicit FMA contraction: the compatibility morph and the Beeston-Barlow roots follow the
           # reference's operation order exactly; the hot loop uses explicit fma()
           '-ffp-contract=off',
           # fp64 MFMA accumulators in VGPRs: the scan kernels compare / use every matrix element with vector
           # instructions, and with the accumulators in AGPRs each element costs two v_accvgpr_read first (no vector
           # instruction executes beside an fp64 MFMA on gfx950, so every one of them is MFMA time lost)
           '-mllvm', '--amdgpu-mfma-vgpr-form',
           '-Wall', '-Wno-unused-function', '-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


HOST_SRC = os.path.join(CSRC, 'host_backend.cpp')
HOST_OUT = os.path.join(OUT_DIR, 'libblueice_host.so')


def build_host(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(HOST_OUT) and os.path.getmtime(HOST_OUT) >= max(os.path.getmtime(HOST_SRC), os.path.getmtime(HDR)):
        return HOST_OUT
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-shared', '-fPIC',
           '-ffp-contract=off',          # every multiplication and addition rounds on its own, as numpy's do
           '-Wall', '-Wextra', '-o', HOST_OUT, HOST_SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return HOST_OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_host(force='--force' in sys.argv, verbose=True))
