"""ctypes loader for oracle/lsap.c (test infrastructure; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_lsap.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_lsap.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        _lib.oracle_lsap.restype = ctypes.c_int
    return _lib


def linear_sum_assignment(cost):
    """Same contract as scipy.optimize.linear_sum_assignment (minimisation only).

    Restates the call at /root/reference/models/matcher.py:85."""
    c = np.ascontiguousarray(np.asarray(cost), dtype=np.float64)
    if c.ndim != 2:
        raise ValueError("expected a matrix (2-D array), got a %d array" % c.ndim)
    nr, nc = c.shape
    n = min(nr, nc)
    rows = np.zeros(n, dtype=np.int64)
    cols = np.zeros(n, dtype=np.int64)
    rc = _load().oracle_lsap(c.ctypes.data, nr, nc, rows.ctypes.data, cols.ctypes.data)
    if rc == -2:
        raise ValueError("matrix contains invalid numeric entries")
    if rc == -1:
        raise ValueError("cost matrix is infeasible")
    if rc != 0:
        raise MemoryError("oracle_lsap failed")
    return rows, cols
