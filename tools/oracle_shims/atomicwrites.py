"""Minimal stand-in for the third-party `atomicwrites` package (absent from this image).

Build-authored code, NOT reference code; used only by tests/golden/make_golden.py in the
development container (call site in the reference: blueice/utils.py:76).
"""
import os
import tempfile
from contextlib import contextmanager


@contextmanager
def atomic_write(path, mode="w", overwrite=False, **kwargs):
    d = os.path.dirname(os.path.abspath(path))
    fd, tmp = tempfile.mkstemp(dir=d)
    os.close(fd)
    try:
        with open(tmp, mode) as f:
            yield f
        if not overwrite and os.path.exists(path):
            raise FileExistsError(path)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
