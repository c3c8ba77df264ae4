"""Build libblueice_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU): six translation units compiled
side by side, rebuilt when the sha256 of the sources' content changes (lib/libblueice_hip.sha256).

    python -m blueice_amd.build [--force]

build_host() compiles csrc/host_backend.cpp -- the minimal entry points of the same C ABI as plain C++ loops on the
host -- into lib/libblueice_host.so.  That library is a BOUNDARY TEST BUILD: the package
never loads it (blueice_amd/_capi.py binds libblueice_hip.so only); tests and tools/ load it by explicit path.
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# one translation unit per heavy kernel family (csrc/bi_common.h says which), compiled side by side
UNITS = ('blueice_hip', 'tu_morph', 'tu_scan', 'tu_scan_sorted', 'tu_grad', 'tu_scan_bb', 'tu_prim')
HDR = os.path.join(os.path.dirname(_HERE), 'include', 'blueice_hip.h')
OUT_DIR = os.path.join(_HERE, 'lib')
OBJ_DIR = os.path.join(_HERE, 'lib', 'obj')
OUT = os.path.join(OUT_DIR, 'libblueice_hip.so')
STAMP = os.path.join(OUT_DIR, 'libblueice_hip.sha256')
ARCH = 'gfx950'

FLAGS = ['-O3', '-std=c++17', '--offload-arch=' + ARCH, '-fPIC',
         # no implicit FMA contraction: the compatibility morph and the Beeston-Barlow roots follow the
         # reference's operation order exactly; the hot loop uses explicit fma()
         '-ffp-contract=off',
         # fp64 MFMA accumulators in VGPRs: the scan kernels compare / use every matrix element with vector
         # instructions, and with the accumulators in AGPRs each element costs two v_accvgpr_read first (no vector
         # instruction executes beside an fp64 MFMA on gfx950, so every one of them is MFMA time lost)
         '-mllvm', '--amdgpu-mfma-vgpr-form',
         # only the C ABI of include/blueice_hip.h is exported; the launchers between the translation units stay inside
         '-fvisibility=hidden',
         '-Wall', '-Wno-unused-function']
# kernel experiments (tools/tune_*): extra compiler arguments, part of the content hash like the others
FLAGS += os.environ.get('BLUEICE_AMD_EXTRA_FLAGS', '').split()


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return 'hipcc'


_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path, seen=None):
    """the file and every file it includes with quotes, recursively (the library's own sources; <...> headers are the toolchain's)"""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as fh:
        text = fh.read()
    for inc in _INCLUDE.findall(text):
        _closure(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def unit_hash(name):
    """sha256 over the CONTENT of a translation unit and of everything it includes, and the flags: what its rebuild is gated
    on (a fresh clone has fresh mtimes but the same content; a touched file with the same bytes is not a change)."""
    h = hashlib.sha256()
    h.update(' '.join(FLAGS).encode())
    for f in sorted(_closure(os.path.join(CSRC, name + '.hip'))):
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def source_hash():
    h = hashlib.sha256()
    for u in UNITS:
        h.update(unit_hash(u).encode())
    return h.hexdigest()


def up_to_date():
    if not (os.path.exists(OUT) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == source_hash()


def build(force=False, verbose=False, jobs=None):
    os.makedirs(OBJ_DIR, exist_ok=True)
    if not force and up_to_date():
        return OUT
    digest = source_hash()
    cc = hipcc()

    def compile_unit(name):
        obj = os.path.join(OBJ_DIR, name + '.o')
        stamp = obj + '.sha256'
        want = unit_hash(name)
        if not force and os.path.exists(obj) and os.path.exists(stamp):
            with open(stamp) as fh:
                if fh.read().strip() == want:
                    return obj                      # this unit's sources have not changed
        cmd = [cc] + FLAGS + ['-c', '-o', obj, os.path.join(CSRC, name + '.hip')]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, 'w') as fh:
            fh.write(want + '\n')
        return obj

    jobs = jobs or max(1, min(len(UNITS), (os.cpu_count() or 2)))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_unit, UNITS))
    cmd = [cc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-fvisibility=hidden', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, 'w') as fh:
        fh.write(digest + '\n')
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
