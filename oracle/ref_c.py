"""ctypes loader of oracle/libqp_ref.so (C restatement of the reference's serial CSC
Chebyshev step).  TEST / BASELINE INFRASTRUCTURE ONLY -- see qp_oracle.py's header."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqp_ref.so")
_lib = None


def usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()) + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return n


def omp_threads():
    """Thread count of the OpenMP variant (fixed before libgomp starts: OMP_NUM_THREADS)."""
    if "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(usable_cores())
    return int(os.environ["OMP_NUM_THREADS"])


def _cpu_signature():
    """Model name + ISA flags of this host: the .so is built with -march=native."""
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        model = next((ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("model name")), "?")
        flags = next((ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("flags")), "?")
        import hashlib
        return model + " / " + hashlib.sha1(flags.encode()).hexdigest()[:12]
    except OSError:
        return "unknown"


def build(force=False):
    """Compile libqp_ref.so for THIS host's CPU (-march=native) and record which CPU that was; if the
    native build fails (e.g. a gcc that does not know the CPU), fall back to the portable x86-64-v2."""
    if force and os.path.exists(_SO):
        os.remove(_SO)
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    except subprocess.CalledProcessError:
        subprocess.check_call(["make", "-s", "-C", _HERE, "MARCH=x86-64-v2"])
    with open(_SO + ".cpu", "w") as f:
        f.write(_cpu_signature())


def build_flags():
    try:
        with open(_SO + ".flags") as f:
            return f.read().strip()
    except OSError:
        return "unknown"


def load():
    global _lib
    if _lib is None:
        omp_threads()
        import fcntl
        # several processes may get here at once (the ranks of a multi-process test on a fresh box): the check, a
        # rebuild and the dlopen happen under one exclusive file lock, so nobody maps a half-written library
        with open(_SO + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            built_on = None
            if os.path.exists(_SO + ".cpu"):
                with open(_SO + ".cpu") as f:
                    built_on = f.read()
            if not os.path.exists(_SO) or built_on != _cpu_signature():
                build(force=True)      # first use on this CPU (the GPU box differs from the build container)
            _lib = C.CDLL(_SO)
        _lib.qp_ref_omp_setup.restype = C.c_void_p
        _lib.qp_ref_omp_setup.argtypes = [C.c_int64] + [C.c_void_p] * 4
        _lib.qp_ref_omp_step.restype = C.c_int
        _lib.qp_ref_omp_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.qp_ref_omp_get_psi.restype = None
        _lib.qp_ref_omp_get_psi.argtypes = [C.c_void_p, C.c_void_p]
        _lib.qp_ref_omp_free.restype = None
        _lib.qp_ref_omp_free.argtypes = [C.c_void_p]
        _lib.qp_ref_cheby_csc.restype = C.c_int
        _lib.qp_ref_cheby_csc.argtypes = [C.c_int64] + [C.c_void_p] * 8 + [C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.qp_ref_cheby_csr_omp.restype = C.c_int
        _lib.qp_ref_cheby_csr_omp.argtypes = [C.c_int64] + [C.c_void_p] * 7 + [C.c_int, C.c_double, C.c_double, C.c_double]
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


def cheby_csr_omp(rowptr, col, val, psi, a, Delta, E_min, dt):
    """All-cores variant (OpenMP, row-parallel CSR, fused passes); in-place on ``psi``.  The
    thread count is OpenMP's (OMP_NUM_THREADS or all cores)."""
    lib = load()
    n = len(psi)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.complex128)
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous
    w = [np.empty(n, dtype=np.complex128) for _ in range(2)]
    p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    nmv = lib.qp_ref_cheby_csr_omp(n, p(rowptr), p(col), p(val), p(psi), p(w[0]), p(w[1]), p(a), len(a),
                                   float(Delta), float(E_min), float(dt))
    assert nmv == len(a) - 1
    return psi


class ChebyCsrOmp:
    """The all-cores variant on buffers first touched in parallel (see cheby_ref.c): set up once,
    ``step()`` advances the internal state by one cheby! call, ``psi()`` copies it out."""

    def __init__(self, rowptr, col, val, psi):
        lib = load()
        self._lib = lib
        n = len(psi)
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.complex128)
        psi = np.ascontiguousarray(psi, dtype=np.complex128)
        p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        self.n = n
        self._h = lib.qp_ref_omp_setup(n, p(rowptr), p(col), p(val), p(psi))
        if not self._h:
            raise MemoryError("qp_ref_omp_setup")

    def step(self, a, Delta, E_min, dt):
        a = np.ascontiguousarray(a, dtype=np.float64)
        nmv = self._lib.qp_ref_omp_step(self._h, a.ctypes.data_as(C.c_void_p), len(a), float(Delta), float(E_min), float(dt))
        assert nmv == len(a) - 1

    def psi(self):
        out = np.empty(self.n, dtype=np.complex128)
        self._lib.qp_ref_omp_get_psi(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    def close(self):
        if self._h:
            self._lib.qp_ref_omp_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
