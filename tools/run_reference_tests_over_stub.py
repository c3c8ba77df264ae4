#!/usr/bin/env python3
"""Development container only: run the REFERENCE'S OWN test files with blueice.likelihood.BinnedLogLikelihood replaced by
the binding of tools/reference_stub/hip_backend.py, i.e. with every binned likelihood evaluation going through the C ABI of
include/blueice_hip.h ("Next round" item 2).  There is no GPU here, so the library behind the ABI is the
host build of the same entry points (blueice_amd/csrc/host_backend.cpp -> blueice_amd/lib/libblueice_host.so); on a machine
with both a GPU and blueice, BLUEICE_HIP_LIB=.../libblueice_hip.so runs the same thing on the device.

    python tools/run_reference_tests_over_stub.py [--out profiles/r04_reference_tests_over_stub.txt]

Nothing of the reference is copied or travels: its tests are run where they lie (/root/reference/tests), from a scratch
directory (sources write ./pdf_cache).  The two third-party modules the image lacks (multihist, atomicwrites) are the
build-authored stand-ins of tools/oracle_shims, as for the golden fixtures."""
import argparse
import io
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('BLUEICE_REFERENCE', '/root/reference')
TEST_FILES = ['test_binned_likelihood.py', 'test_BeestonBarlow.py', 'test_morphers.py', 'test_likelihood.py',
              'test_inference.py']


class Tee(io.TextIOBase):
    def __init__(self, *streams):
        self.streams = streams

    def write(self, s):
        for st in self.streams:
            st.write(s)
        return len(s)

    def flush(self):
        for st in self.streams:
            st.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--lib', default=None, help='library behind the ABI (default: the host build)')
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(REFERENCE, 'blueice')):
        sys.exit('the reference is not here (%s): this script is for the development container' % REFERENCE)
    sys.path.insert(0, ROOT)
    from blueice_amd import build
    lib = args.lib or build.build_host()
    os.environ['BLUEICE_HIP_LIB'] = lib
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(ROOT, 'tools', 'oracle_shims'), REFERENCE, os.path.join(ROOT, 'tools', 'reference_stub')]
    scratch = tempfile.mkdtemp(prefix='blueice_stub_')
    os.chdir(scratch)

    buf = io.StringIO()
    out = Tee(sys.stdout, buf)
    import blueice
    import blueice.likelihood
    import hip_backend
    original = blueice.likelihood.BinnedLogLikelihood
    blueice.likelihood.BinnedLogLikelihood = hip_backend.HipBinnedLogLikelihood        # before the test modules import it
    print('reference: blueice %s at %s' % (blueice.__version__, os.path.dirname(blueice.__file__)), file=out)
    print('binding:   %s (class %s over %s)' % (hip_backend.__file__, hip_backend.HipBinnedLogLikelihood.__name__,
                                               original.__module__ + '.' + original.__name__), file=out)
    print('library:   %s\n           %s' % (lib, hip_backend.library_version()), file=out)
    print('tests:     %s\n' % ' '.join(os.path.join(REFERENCE, 'tests', t) for t in TEST_FILES), file=out)

    import pytest
    from _pytest.config import ExitCode

    class Capture:
        """pytest writes its report through the terminal reporter: run it in-process and keep a copy"""
        def pytest_terminal_summary(self, terminalreporter):
            pass

    stdout = sys.stdout
    sys.stdout = out
    try:
        rc = pytest.main(['-p', 'no:cacheprovider', '-v', '--color=no', '-W', 'ignore'] +
                         [os.path.join(REFERENCE, 'tests', t) for t in TEST_FILES], plugins=[Capture()])
    finally:
        sys.stdout = stdout
    print('\nlibrary calls made by these tests through the binding: %s' % ', '.join('%s x %d' % kv for kv in hip_backend.CALLS.items()),
          file=out)
    print('exit code: %d (%s)' % (int(rc), ExitCode(rc).name), file=out)

    # Second part: the 18 binned likelihoods of tests/model_zoo.py (the builders behind the golden fixtures), built from the
    # reference's classes with the binding in place, every recorded call replayed through bi_eval / bi_eval_full and held
    # against what the UNPATCHED reference returned when the fixtures were made (tests/golden/*.npz) -- bit for bit.
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    import model_zoo
    from golden_util import load_case, same
    ns = model_zoo.namespace_of('blueice')
    assert ns.BinnedLogLikelihood is hip_backend.HipBinnedLogLikelihood
    print('\ngolden fixtures replayed through the binding (reference classes + %s):' % os.path.basename(lib), file=out)
    bad = 0
    before = dict(hip_backend.CALLS)
    stderr = sys.stderr
    sys.stderr = io.StringIO()                       # (progress bars of the reference's model building)
    try:
        for name, builder in model_zoo.CASES.items():
            lf, calls, full = builder(ns)
            gold = load_case(name)
            n_ok = 0
            for j, kw in enumerate(calls):
                asserted = False
                try:
                    ll = lf(**dict(kw))
                except AssertionError:
                    asserted, ll = True, float('nan')
                want_assert = ('call_asserts_%d' % j) in gold['raw'].files
                ok = asserted == want_assert and (asserted or same(ll, gold['call_ll'][j], rtol=0.0))
                if ok and j in full and np.isfinite(ll):
                    r, m, p = lf(full_output=True, **dict(kw))
                    ok = np.array_equal(m, gold['raw']['full_%d_mus' % j]) and np.array_equal(p, gold['raw']['full_%d_ps' % j])
                n_ok += bool(ok)
                if not ok:
                    print('    %s call %d: %r, the reference gave %r' % (name, j, ll, float(gold['call_ll'][j])), file=out)
            bad += len(calls) - n_ok
            print('  %-26s %2d / %2d calls identical' % (name, n_ok, len(calls)), file=out)
    finally:
        sys.stderr = stderr
    print('library calls of this part: %s' % ', '.join('%s x %d' % (k, v - before[k]) for k, v in hip_backend.CALLS.items()), file=out)
    print('fixtures: %s' % ('all identical' if bad == 0 else '%d calls DIFFER' % bad), file=out)
    rc = int(rc) or (1 if bad else 0)
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, 'w') as f:
            f.write(buf.getvalue())
    return int(rc)


if __name__ == '__main__':
    sys.exit(main())
