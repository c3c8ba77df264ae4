"""Development container only (skipped wherever the reference is absent, e.g. on the GPU box): the reference's own test
files and the golden-fixture builders run with blueice.likelihood.BinnedLogLikelihood replaced by the ctypes binding of
tools/reference_stub/hip_backend.py over the host build of the C ABI (tools/run_reference_tests_over_stub.py; the committed
transcript is profiles/r04_reference_tests_over_stub.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('BLUEICE_REFERENCE', '/root/reference')

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'blueice')), reason='the reference is not on this machine')


def test_reference_tests_and_goldens_pass_through_the_binding():
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_reference_tests_over_stub.py')],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    out = res.stdout
    assert res.returncode == 0, out[-4000:] + res.stderr[-2000:]
    assert '21 passed' in out and ' failed' not in out
    for t in ('test_single_bin', 'test_twobin_mc', 'test_multi_bin_single_dim', 'test_multi_bin', 'test_BeestonBarlowSingleBin',
              'test_BeestonBarlowMultiBin', 'test_BeestonBarlow_second_source', 'test_morpher_api', 'test_zero_bin'):
        assert '::%s PASSED' % t in out, t
    assert 'fixtures: all identical' in out
    assert 'libblueice_host.so' in out
