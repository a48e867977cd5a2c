"""ctypes loader of oracle/libqp_ref.so (C restatement of the reference's serial CSC Chebyshev step,
cheby_ref.c, and of the hot loops of its Newton / restarted-Arnoldi step, newton_ref.c).
TEST / BASELINE INFRASTRUCTURE ONLY -- see qp_oracle.py's header."""
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
        _lib.qp_ref_norm.restype = C.c_double
        _lib.qp_ref_norm.argtypes = [C.c_int64, C.c_void_p]
        _lib.qp_ref_scale.restype = None
        _lib.qp_ref_scale.argtypes = [C.c_int64, C.c_void_p, C.c_double, C.c_double]
        _lib.qp_ref_arnoldi_csc.restype = C.c_int
        _lib.qp_ref_arnoldi_csc.argtypes = [C.c_int64] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_double,
                                            C.c_void_p, C.c_int, C.c_void_p]
        _lib.qp_ref_lincomb.restype = None
        _lib.qp_ref_lincomb.argtypes = [C.c_int64, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
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


def _p(x):
    return x.ctypes.data_as(C.c_void_p)


def arnoldi_csc(colptr, rowval, nzval, q, m, psi, dt=1.0, extended=True, norm_min=1e-15):
    """``arnoldi!`` (src/arnoldi.jl:60-100) in C on a CSC matrix: ``q`` is a C-contiguous ((>= m + 1), n) complex128 array
    (row k = basis vector k), filled in place.  -> (effective m, Hess as an (ld, ld) array with Hess[i, j] = Julia's
    Hess[i+1, j+1], ld = m + 1)."""
    lib = load()
    n = q.shape[1]
    assert q.dtype == np.complex128 and q.flags.c_contiguous and q.shape[0] >= m + 1
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = np.ascontiguousarray(nzval, dtype=np.complex128)
    psi = np.ascontiguousarray(psi, dtype=np.complex128)
    assert len(colptr) == n + 1 and len(psi) == n
    ld = m + 1
    hess_f = np.zeros(ld * ld, dtype=np.complex128)              # column-major, as the Julia matrix
    nmv = C.c_int(0)
    m_eff = lib.qp_ref_arnoldi_csc(n, _p(colptr), _p(rowval), _p(nzval), _p(q), int(m), _p(psi), float(dt), int(bool(extended)),
                                   float(norm_min), _p(hess_f), ld, C.byref(nmv))
    assert m_eff >= 1
    return m_eff, hess_f.reshape(ld, ld).T.copy(), int(nmv.value)


class NewtonCscWrk:
    """``NewtonWrk`` (src/newton.jl:23-60) for :func:`newton_csc`: the basis as one contiguous array."""

    def __init__(self, n, m_max=10):
        if m_max >= n:
            m_max = n - 1
        assert m_max > 2
        self.m_max = m_max
        self.q = np.zeros((m_max + 1, n), dtype=np.complex128)
        self.v = np.zeros(n, dtype=np.complex128)
        self.a = np.zeros(10 * m_max + 1, dtype=np.complex128)
        self.leja = np.zeros(10 * m_max + 1, dtype=np.complex128)
        self.radius = 0.0
        self.restarts = self.n_a = self.n_leja = self.n_matvec = 0


def newton_csc(colptr, rowval, nzval, psi, dt, wrk, func=None, norm_min=1e-14, relerr=1e-12, max_restarts=50):
    """In-place ``newton!`` (src/newton.jl:246-385) on a CSC matrix (0-based int64 / complex128) with every O(N) operation
    in C (newton_ref.c: serial CSC mat-vec, sequential modified Gram-Schmidt, the two combinations, norms) and the small host
    algebra between them -- Ritz values of the Hessenberg blocks, Leja ordering, divided differences: a few hundred scalars
    per restart -- by the NumPy oracle's own functions (qp_oracle.py, pinned by tests/test_oracle_kat.py).  The statement
    order is that of qp_oracle.newton, which follows the reference line by line."""
    import importlib.util
    import sys
    qo = sys.modules.get("oracle.qp_oracle") or sys.modules.get("qp_oracle")
    if qo is None:
        spec = importlib.util.spec_from_file_location("qp_oracle", os.path.join(_HERE, "qp_oracle.py"))
        qo = importlib.util.module_from_spec(spec)
        sys.modules["qp_oracle"] = qo
        spec.loader.exec_module(qo)
    lib = load()
    func = qo._expmi if func is None else func
    n = len(psi)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = np.ascontiguousarray(nzval, dtype=np.complex128)
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous
    m = wrk.m_max
    wrk.a[:] = 0                                                 # :254
    wrk.leja[:] = 0                                              # :255
    _dt = float(dt)
    assert _dt != 0.0
    n_a = n_leja = 0
    wrk.v[:] = psi                                               # :268
    s = 0
    beta = lib.qp_ref_norm(n, _p(wrk.v))                         # :271
    lib.qp_ref_scale(n, _p(wrk.v), 1.0 / beta, 0.0)              # :272
    q = wrk.q
    while True:                                                  # :274
        m, Hess, nmv = arnoldi_csc(colptr, rowval, nzval, q, m, wrk.v, _dt, True, norm_min)      # :277
        wrk.n_matvec += nmv
        if m == 1 and s == 0:                                    # :289-295
            lam = complex(func(beta * Hess[0, 0]))
            lib.qp_ref_scale(n, _p(psi), lam.real, lam.imag)
            break
        ritz = qo.diagonalize_hessenberg_matrix(Hess, m, accumulate=True)                       # :297
        if s == 0:                                               # :301-303
            wrk.radius = 1.2 * np.max(np.abs(ritz))
        n_s = n_leja                                             # :307
        wrk.leja, n_leja = qo.extend_leja(wrk.leja, n_leja, ritz.copy(), m)
        wrk.a, n_a = qo.extend_newton_coeffs(wrk.a, n_a, wrk.leja, func, n_leja, wrk.radius)     # :314
        assert n_a == n_leja
        R = np.zeros(m + 1, dtype=np.complex128)                 # :330-333
        P = np.zeros(m + 1, dtype=np.complex128)
        R[0] = beta
        P[0] = wrk.a[n_s] * beta
        Hm = Hess[:m + 1, :m + 1]
        for k in range(1, m):                                    # :334-343
            R = (Hm @ R - wrk.leja[n_s + k - 1] * R) / wrk.radius
            P = P + wrk.a[n_s + k] * R
        Pm = np.ascontiguousarray(P[:m])
        lib.qp_ref_lincomb(n, _p(psi), 0.0 if s == 0 else 1.0, 0.0, _p(q), _p(Pm), m)            # :346-352
        R = (Hm @ R - wrk.leja[n_s + m - 1] * R) / wrk.radius    # :356-359
        beta = float(np.linalg.norm(np.abs(R)))                  # :360-361
        R = R * (1 / beta)                                       # :362
        q[0, :] = wrk.v                                          # :363
        Rt = np.ascontiguousarray(R[1:m + 1])
        lib.qp_ref_lincomb(n, _p(wrk.v), R[0].real, R[0].imag, _p(q[1:]), _p(Rt), m)             # :364-367
        norm_psi = lib.qp_ref_norm(n, _p(psi))
        if beta * abs(wrk.a[n_a - 1]) / (1 + norm_psi) < relerr:  # :370
            break
        s += 1
        assert s <= max_restarts, "max_restarts exceeded"        # :375
    wrk.restarts, wrk.n_leja, wrk.n_a = s, n_leja, n_a
    return psi
