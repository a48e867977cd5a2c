"""ctypes loader of oracle/libqp_ref.so (C restatement of the reference's serial CSC
Chebyshev step).  TEST / BASELINE INFRASTRUCTURE ONLY -- see qp_oracle.py's header."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqp_ref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.qp_ref_cheby_csc.restype = C.c_int
        _lib.qp_ref_cheby_csc.argtypes = [C.c_int64] + [C.c_void_p] * 8 + [C.c_int, C.c_double, C.c_double, C.c_double]
    return _lib


def cheby_csc(colptr, rowval, nzval, psi, a, Delta, E_min, dt):
    """In-place on ``psi`` (complex128).  CSC arrays 0-based int64 / complex128."""
    lib = load()
    n = len(psi)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = np.ascontiguousarray(nzval, dtype=np.complex128)
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous
    w = [np.empty(n, dtype=np.complex128) for _ in range(3)]
    p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    nmv = lib.qp_ref_cheby_csc(n, p(colptr), p(rowval), p(nzval), p(psi), p(w[0]), p(w[1]), p(w[2]), p(a),
                               len(a), float(Delta), float(E_min), float(dt))
    assert nmv == len(a) - 1
    return psi
