"""CPU oracle: NumPy restatement of the QuantumPropagators.jl `prop_step!` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the *checker*.  The product path
(``quantumpropagators.jl_amd``) never imports this module and fails loudly when
its HIP library is missing.

Every function cites the reference file:line (paths relative to the reference
checkout of JuliaQuantumControl/QuantumPropagators.jl, v0.8.5+dev) it restates.

Pinning status
--------------
The reference is pure Julia; ``julia`` is not installed in the build container,
so the reference can be neither run nor imported, and its tests hold *no stored
golden vectors* for this path -- they are self-checking known-answer tests
(dense ``exp``, ``eigvals``, analytic two-level results, Newton-vs-Cheby on the
deterministic optomechanics Hamiltonian).  This oracle is therefore pinned by
restating exactly those known-answer tests (``tests/test_oracle_kat.py``):
``test/test_cheby.jl:6-49``, ``test/test_newton.jl:7-177``,
``test/test_specrad.jl:14-223``, ``test/test_propagate.jl:74-163`` with
``test/optomech.jl``, ``test/test_operator_linalg.jl:30-64``.

Conventions: states are 1-D ``complex128`` arrays; operators are anything with
``@`` (numpy arrays, scipy.sparse matrices) or :class:`Operator` instances.
Indices that are 1-based in Julia are 0-based here unless noted.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
from scipy.special import jv as _besselj

__all__ = [
    "cheby_coeffs", "ChebyWrk", "cheby", "arnoldi", "extend_arnoldi",
    "diagonalize_hessenberg_matrix", "extend_leja", "extend_newton_coeffs",
    "NewtonWrk", "newton", "ritzvals", "specrange", "Operator",
    "ScaledOperator", "Generator", "matvec", "csc_to_csr", "partition_rows",
    "PWCPropagator", "init_prop", "prop_step", "propagate",
]


# --------------------------------------------------------------------------
# generic mat-vec used by all kernels: mul!(y, H, x)
# --------------------------------------------------------------------------

def matvec(H, x):
    """``mul!(y, H, x)`` -- 3-arg form; for :class:`Operator` the lazy sum of
    src/generators.jl:634-645 with (alpha, beta) = (1, 0)."""
    if isinstance(H, (Operator, ScaledOperator)):
        return H.mul(x)
    return np.asarray(H @ x).reshape(-1)


# --------------------------------------------------------------------------
# Cheby  (src/cheby.jl)
# --------------------------------------------------------------------------

def cheby_coeffs(Delta, dt, limit=1e-12):
    """src/cheby.jl:25-39.  a_1 = J_0(alpha); a_k = 2 J_{k-1}(alpha), appended
    until the just-appended |a_k| <= limit (that last one is kept)."""
    alpha = abs(0.5 * Delta * dt)
    coeffs = []
    a = float(_besselj(0, alpha))
    coeffs.append(a)
    eps = abs(a)
    i = 1
    while eps > limit:
        a = 2.0 * float(_besselj(i, alpha))
        coeffs.append(a)
        eps = abs(a)
        i += 1
    return np.array(coeffs, dtype=np.float64)


class ChebyWrk:
    """src/cheby.jl:87-124."""

    def __init__(self, psi, Delta, E_min, dt, limit=1e-12):
        self.v0 = np.empty_like(psi)
        self.v1 = np.empty_like(psi)
        self.v2 = np.empty_like(psi)
        self.coeffs = cheby_coeffs(Delta, dt, limit=limit)
        self.n_coeffs = len(self.coeffs)
        self.Delta = float(Delta)
        self.E_min = float(E_min)
        self.dt = float(dt)
        self.limit = float(limit)
        self.n_matvec = 0


def cheby(psi, H, dt, wrk, E_min=None, check_normalization=False):
    """In-place ``cheby!`` -- src/cheby.jl:150-213.  ``psi`` is overwritten and
    returned.  The sequence of BLAS-1 operations is kept exactly."""
    E_min = wrk.E_min if E_min is None else E_min
    Delta = wrk.Delta
    beta = (Delta / 2) + E_min
    # src/cheby.jl:157
    assert math.isclose(abs(dt), abs(wrk.dt), rel_tol=math.sqrt(np.finfo(float).eps)), (
        f"wrk was initialized for dt={wrk.dt}, not dt=abs({dt})")
    c = (-2j / Delta) if dt > 0 else (2j / Delta)
    a = wrk.coeffs
    eps = wrk.limit
    assert len(a) > 1, "Need at least 2 Chebychev coefficients"
    v0, v1, v2 = wrk.v0, wrk.v1, wrk.v2

    v0[:] = psi                  # copyto!(v0, Psi)               :171
    psi *= a[0]                  # lmul!(a[1], Psi)               :172
    v1[:] = matvec(H, v0)        # mul!(v1, H, v0)                :176
    wrk.n_matvec += 1
    v1 += (-beta) * v0           # axpy!(-beta, v0, v1)           :178
    v1 *= c                      # lmul!(c, v1)                   :179
    psi += a[1] * v1             # axpy!(a[2], v1, Psi)           :182
    c *= 2                       #                                 :184
    for i in range(2, wrk.n_coeffs):                            # :186
        v2[:] = matvec(H, v1)    #                                 :190
        wrk.n_matvec += 1
        v2 += (-beta) * v1       #                                 :192
        v2 *= c                  #                                 :193
        if check_normalization:  #                                 :194-200
            map_norm = abs(np.vdot(v1, v2)) / (2 * np.linalg.norm(v1) ** 2)
            assert map_norm <= (1.0 + eps), (
                f"Incorrect normalization (E_min={E_min}, Delta={Delta})")
        v2 += v0                 #                                 :202
        psi += a[i] * v2         #                                 :205
        v0, v1, v2 = v1, v2, v0  #                                 :207
    psi *= np.exp(-1j * beta * dt)                              # :211
    return psi


# --------------------------------------------------------------------------
# Arnoldi (src/arnoldi.jl)
# --------------------------------------------------------------------------

def arnoldi(Hess, q, m, psi, H, dt=1.0, extended=True, norm_min=1e-15):
    """``arnoldi!`` -- src/arnoldi.jl:60-100.  ``Hess`` is a 2-D complex array
    (zero-filled here), ``q`` a list of >= m+1 pre-allocated vectors.  Returns
    the effective m.  Hess[i, j] (0-based) == Julia Hess[i+1, j+1]."""
    dim_hess = m + 1 if extended else m
    assert Hess.shape[0] >= dim_hess and Hess.shape[1] >= dim_hess
    assert len(q) >= m + 1
    Hess[:] = 0                                                  # :78
    q[0][:] = psi                                                # :79
    for j in range(m):                                           # :80
        q[j + 1][:] = matvec(H, q[j])                            # :82
        for i in range(j + 1):                                   # :84
            Hess[i, j] = dt * np.vdot(q[i], q[j + 1])            # :85
            q[j + 1] += (-Hess[i, j] / dt) * q[i]                # :86
        if (j + 1 < m) or extended:                              # :88
            h = np.linalg.norm(q[j + 1])
            Hess[j + 1, j] = dt * h
            if h < norm_min:                                     # :91
                m = j + 1
                break
            q[j + 1] *= (1 / h)                                  # :96
    return m


def extend_arnoldi(Hess, q, m, H, dt=1.0, norm_min=1e-15):
    """``extend_arnoldi!`` -- src/arnoldi.jl:115-129, with Julia's 1-based m:
    extends Hess from (m-1)x(m-1) to m x m and q from m to m+1 vectors.
    Returns True when the extension was done, False on the early return at
    :117 (callers in the reference ignore the return value)."""
    h = np.linalg.norm(q[m - 1])                                 # :116
    if h < norm_min:                                             # :117
        return False
    Hess[m - 1, m - 2] = dt * h                                  # :118
    q[m - 1] *= (1 / h)                                          # :119
    q[m][:] = matvec(H, q[m - 1])                                # :120
    for i in range(m):                                           # :121
        Hess[i, m - 1] = dt * np.vdot(q[i], q[m])                # :122
        q[m] += (-Hess[i, m - 1] / dt) * q[i]                    # :123
    assert np.all(Hess[m - 1, 0:(m - 2)] == 0.0)                 # :127
    return True


def _eigvals_sorted(A):
    """Julia ``eigvals`` of a general complex matrix: LAPACK values sorted by
    (real, imag) (LinearAlgebra default ``sortby``)."""
    ev = np.linalg.eigvals(A)
    order = np.lexsort((ev.imag, ev.real))
    return ev[order]


def diagonalize_hessenberg_matrix(Hess, m, accumulate=False):
    """src/arnoldi.jl:143-170."""
    j_min = 1 if accumulate else m
    n_out = (m * (m + 1)) // 2 if accumulate else m
    eigenvals = np.zeros(n_out, dtype=np.complex128)
    offset = 0
    for j in range(j_min, m + 1):
        if j == 1:
            eigenvals[0] = Hess[0, 0]
        elif j == 2:
            a, c, b, d = Hess[0, 0], Hess[1, 0], Hess[0, 1], Hess[1, 1]
            s = np.sqrt(complex(a * a + 4 * b * c - 2 * a * d + d * d))
            eigenvals[offset + 0] = 0.5 * (a + d - s)
            eigenvals[offset + 1] = 0.5 * (a + d + s)
        else:
            eigenvals[offset:offset + j] = _eigvals_sorted(Hess[:j, :j])
        offset += j
    return eigenvals


# --------------------------------------------------------------------------
# Newton (src/newton.jl)
# --------------------------------------------------------------------------

def extend_leja(leja, n, newpoints, n_use):
    """``extend_leja!`` -- src/newton.jl:97-148.  ``leja`` and ``newpoints``
    are zero-based complex arrays (as in the reference, which uses
    OffsetArrays).  ``newpoints`` is clobbered.  Returns (leja, n + n_use);
    ``leja`` may be a re-allocated array."""
    if len(leja) < n + n_use:                                    # :105-110
        new = np.zeros(2 * (n + n_use), dtype=np.complex128)
        new[:n] = leja[:n]
        leja = new
    u = len(newpoints) - 1
    i_add_start = 0
    if n == 0:                                                   # :113-126
        z_last = newpoints[u]
        for i in range(0, u):
            if abs(newpoints[i]) > abs(z_last):
                newpoints[u] = newpoints[i]
                newpoints[i] = z_last
                z_last = newpoints[u]
        leja[0] = newpoints[-1]
        i_add_start = 1
    exponent = 1.0 / (n + n_use)                                 # :127
    for i_add in range(i_add_start, n_use):                      # :128
        p_max = 0.0
        i_max = 0
        for i in range(0, u - i_add + 1):                        # :131
            p = 1.0
            for j in range(0, n + i_add):                        # :133
                d = abs(newpoints[i] - leja[j])
                p = p * d ** exponent
            if p > p_max:                                        # :137 strict
                p_max = p
                i_max = i
        leja[n + i_add] = newpoints[i_max]                       # :143
        newpoints[i_max] = newpoints[u - i_add]                  # :145
    return leja, n + n_use


def extend_newton_coeffs(a, n_a, leja, func, n_leja, radius):
    """``extend_newton_coeffs!`` -- src/newton.jl:176-214 (zero-based a, leja).
    Returns (a, n_a_new)."""
    m = n_leja - n_a
    n0 = n_a
    if len(a) < n_a + m:                                         # :187-192
        new = np.zeros(2 * n_leja, dtype=np.complex128)
        new[:n_a] = a[:n_a]
        a = new
    assert len(leja) >= n_leja
    assert radius > 0
    if n_a == 0:                                                 # :195-198
        a[0] = func(leja[0])
        n0 = 1
    for k in range(n0, n_a + m):                                 # :199
        d = 1.0 + 0j
        pn = 0.0 + 0j
        for n in range(1, k):                                    # :202
            zd = leja[k] - leja[n - 1]
            d = d * zd / radius
            pn = pn + a[n] * d
        zd = leja[k] - leja[k - 1]
        d = d * zd / radius
        assert abs(d) > 1e-200, "Divided differences too small"  # :209
        a[k] = (func(leja[k]) - a[0] - pn) / d                   # :210
    return a, n_a + m


class NewtonWrk:
    """src/newton.jl:23-60."""

    def __init__(self, v0, m_max=10):
        if m_max <= 2:
            raise ValueError("Newton propagation requires m_max > 2")
        if m_max >= len(v0):
            m_max = len(v0) - 1
            if m_max <= 2:
                raise ValueError("Newton propagation requires state dimension > 2")
        self.arnoldi_vecs = [np.empty_like(v0) for _ in range(m_max + 1)]
        self.v = np.empty_like(v0)
        self.a = np.zeros(10 * m_max + 1, dtype=np.complex128)
        self.leja = np.zeros(10 * m_max + 1, dtype=np.complex128)
        self.radius = 0.0
        self.n_a = 0
        self.n_leja = 0
        self.restarts = 0
        self.m_max = m_max
        self.n_matvec = 0
        self.trace = []  # per-restart intermediates for golden fixtures


def _expmi(z):
    return np.exp(-1j * z)


def newton(psi, H, dt, wrk, func=None, norm_min=1e-14, relerr=1e-12,
           max_restarts=50, record=False):
    """In-place ``newton!`` -- src/newton.jl:246-385."""
    func = _expmi if func is None else func
    m_max = len(wrk.arnoldi_vecs) - 1
    m = m_max
    wrk.a[:] = 0                                                 # :254
    wrk.leja[:] = 0                                              # :255
    Hess = np.zeros((m_max + 1, m_max + 1), dtype=np.complex128)
    _dt = float(dt)
    assert _dt != 0.0
    n_a = 0
    n_leja = 0
    wrk.v[:] = psi                                               # :268
    s = 0
    beta = np.linalg.norm(wrk.v)                                 # :271
    wrk.v *= (1 / beta)                                          # :272
    if record:
        wrk.trace = []
    while True:                                                  # :274
        m = arnoldi(Hess, wrk.arnoldi_vecs, m, wrk.v, H, _dt,
                    extended=True, norm_min=norm_min)            # :277
        wrk.n_matvec += m
        if m == 1 and s == 0:                                    # :289
            lam = beta * Hess[0, 0]
            psi *= func(lam)
            break
        ritz = diagonalize_hessenberg_matrix(Hess, m, accumulate=True)  # :297
        if s == 0:                                               # :301
            wrk.radius = 1.2 * np.max(np.abs(ritz))              # :67-70
        n_s = n_leja                                             # :307
        wrk.leja, n_leja = extend_leja(wrk.leja, n_leja, ritz.copy(), m)
        wrk.a, n_a = extend_newton_coeffs(wrk.a, n_a, wrk.leja, func,
                                          n_leja, wrk.radius)    # :314
        assert n_a == n_leja
        R = np.zeros(m + 1, dtype=np.complex128)                 # :330-331
        P = np.zeros(m + 1, dtype=np.complex128)
        R[0] = beta                                              # :332
        P[0] = wrk.a[n_s] * beta                                 # :333
        Hm = Hess[:m + 1, :m + 1]
        for k in range(1, m):                                    # :334
            z = wrk.leja[n_s + k - 1]
            R = (Hm @ R - z * R) / wrk.radius                    # :336-339
            P = P + wrk.a[n_s + k] * R                           # :341
        if s == 0:                                               # :346
            psi[:] = 0
        for i in range(m):                                       # :350
            psi += P[i] * wrk.arnoldi_vecs[i]
        R = (Hm @ R - wrk.leja[n_s + m - 1] * R) / wrk.radius    # :356-359
        beta = np.linalg.norm(np.abs(R))                         # :360-361
        R = R * (1 / beta)                                       # :362
        wrk.arnoldi_vecs[0][:] = wrk.v                           # :363
        wrk.v *= R[0]                                            # :364
        for i in range(1, m + 1):                                # :365
            wrk.v += R[i] * wrk.arnoldi_vecs[i]
        norm_psi = np.linalg.norm(psi)
        if record:
            wrk.trace.append(dict(m=m, Hess=Hess.copy(), ritz=ritz.copy(),
                                  radius=wrk.radius, n_leja=n_leja,
                                  leja=wrk.leja[:n_leja].copy(),
                                  a=wrk.a[:n_a].copy(), P=P.copy(), R=R.copy(),
                                  beta=beta, norm_psi=norm_psi))
        psi_relerr = beta * abs(wrk.a[n_a - 1]) / (1 + norm_psi)  # :370
        if psi_relerr < relerr:
            break
        s += 1
        assert s <= max_restarts, "max_restarts exceeded"        # :375
    wrk.restarts = s
    wrk.n_leja = n_leja
    wrk.n_a = n_a
    return psi


# --------------------------------------------------------------------------
# SpectralRange (src/specrad.jl)
# --------------------------------------------------------------------------

def ritzvals(G, state, m_min, m_max=None, prec=1e-5, norm_min=1e-15):
    """src/specrad.jl:170-220.  Quirk kept: Krylov exhaustion during the
    extension loop is never detected (:204-205)."""
    if m_max is None:
        m_max = 2 * m_min
    if m_max <= m_min:
        raise ValueError(f"m_max={m_max} must be smaller than m_min={m_min}")
    m = max(5, min(m_min, m_max - 1))
    Hess = np.zeros((m_max, m_max), dtype=np.complex128)
    q = [np.empty_like(state) for _ in range(m_max + 1)]
    m0 = m - 1
    m0 = arnoldi(Hess, q, m0, state, G, extended=False, norm_min=norm_min)  # :182
    ev = diagonalize_hessenberg_matrix(Hess, m0)
    vr0_lo = np.min(ev.real)
    vr0_hi = np.max(ev.real)
    vi0 = np.max(np.abs(ev.imag))
    if m0 == m - 1:
        extend_arnoldi(Hess, q, m, G, norm_min=norm_min)         # :190
        ev = diagonalize_hessenberg_matrix(Hess, m)
        vr_lo, vr_hi, vi = np.min(ev.real), np.max(ev.real), np.max(np.abs(ev.imag))
        er_lo = abs(1.0 - vr_lo / vr0_lo) if vr0_lo != 0.0 else 0.0
        er_hi = abs(1.0 - vr_hi / vr0_hi) if vr0_hi != 0.0 else 0.0
        ei = abs(1.0 - vi / vi0) if vi0 != 0.0 else 0.0
        while (er_lo > prec) or (er_hi > prec) or ((vi0 > 1e-14) and ei > prec):
            vr0_lo, vr0_hi, vi0 = vr_lo, vr_hi, vi
            m0 = m
            m = m + 1
            extend_arnoldi(Hess, q, m, G, norm_min=norm_min)     # :204
            ev = diagonalize_hessenberg_matrix(Hess, m)
            vr_lo, vr_hi, vi = np.min(ev.real), np.max(ev.real), np.max(np.abs(ev.imag))
            with np.errstate(divide="ignore", invalid="ignore"):
                er_lo = abs(1.0 - (vr_lo / vr0_lo))
                er_hi = abs(1.0 - (vr_hi / vr0_hi))
                ei = abs(1.0 - (vi / vi0))
            if m == m_max:
                break
    return ev


def specrange(H, method="auto", state=None, **kw):
    """src/specrad.jl:36-140.  ``state`` is mandatory for ``arnoldi`` here
    (the reference draws a random one, :153-158, not reproducible)."""
    if method == "auto":                                         # :45-61
        if "E_min" in kw and "E_max" in kw:
            return specrange(H, "manual", **kw)
        n = H.shape[0]
        if n <= 32:
            return specrange(H, "diag", **kw)
        return specrange(H, "arnoldi", state=state, **kw)
    if method == "manual":                                       # :138-140
        return float(kw["E_min"]), float(kw["E_max"])
    if method == "diag":                                         # :124-128
        A = H.toarray() if hasattr(H, "toarray") else np.asarray(H)
        ev = _eigvals_sorted(A).real
        return float(ev[0]), float(ev[-1])
    if method == "arnoldi":                                      # :88-112
        m_max = kw.get("m_max", 60)
        m_min = max(5, min(kw.get("m_min", 25), m_max - 1))
        prec = kw.get("prec", 1e-3)
        norm_min = kw.get("norm_min", 1e-15)
        enlarge = kw.get("enlarge", True)
        assert state is not None, "oracle specrange(arnoldi) needs an explicit start state"
        R = ritzvals(H, state, m_min, m_max, prec=prec, norm_min=norm_min)
        E_min = R[0].real
        E_max = R[-1].real
        if enlarge and len(R) > 1:
            E_min = 2 * E_min - R[1].real
            E_max = 2 * E_max - R[-2].real
        return float(E_min), float(E_max)
    raise ValueError(f"unknown specrange method {method!r}")


# --------------------------------------------------------------------------
# Generators / Operators (src/generators.jl)
# --------------------------------------------------------------------------

class Operator:
    """Lazy sum  sum_l c_l H_l  with drift terms having c = 1
    (src/generators.jl:111-125)."""

    def __init__(self, ops, coeffs):
        self.ops = list(ops)
        self.coeffs = list(coeffs)
        assert len(self.coeffs) <= len(self.ops)

    @property
    def shape(self):
        return self.ops[0].shape

    def mul(self, B, alpha=1.0, beta=0.0, C=None):
        """``mul!(C, A::Operator, B, alpha, beta)`` -- src/generators.jl:634-645."""
        drift_offset = len(self.ops) - len(self.coeffs)
        c = alpha
        if drift_offset == 0:
            c = c * self.coeffs[0]
        first = c * np.asarray(self.ops[0] @ B).reshape(-1)
        if C is None or beta == 0:
            out = first
        else:
            out = beta * C + first
        for i in range(1, len(self.ops)):
            c = alpha
            if i + 1 > drift_offset:
                c = c * self.coeffs[i - drift_offset]
            out = out + c * np.asarray(self.ops[i] @ B).reshape(-1)
        return out

    def dot(self, x, y):
        """``dot(x, A::Operator, y)`` -- src/generators.jl:648-660."""
        drift_offset = len(self.ops) - len(self.coeffs)
        result = 0j
        for i, op in enumerate(self.ops):
            d = np.vdot(x, np.asarray(op @ y).reshape(-1))
            if i + 1 > drift_offset:
                result += self.coeffs[i - drift_offset] * d
            else:
                result += d
        return result

    def toarray(self):
        drift_offset = len(self.ops) - len(self.coeffs)
        A = 0
        for i, op in enumerate(self.ops):
            M = op.toarray() if hasattr(op, "toarray") else np.asarray(op)
            c = self.coeffs[i - drift_offset] if i + 1 > drift_offset else 1.0
            A = A + c * M
        return A

    def __matmul__(self, x):
        return self.mul(x)


class ScaledOperator:
    """src/generators.jl:238-249, mul! at :701-703."""

    def __init__(self, coeff, operator):
        self.coeff = coeff
        self.operator = operator

    @property
    def shape(self):
        return self.operator.shape

    def mul(self, B, alpha=1.0, beta=0.0, C=None):
        return self.operator.mul(B, self.coeff * alpha, beta, C)

    def toarray(self):
        return self.coeff * self.operator.toarray()

    def __matmul__(self, x):
        return self.mul(x)


class Generator:
    """sum_l a_l(t) H_l  (src/generators.jl:44-61).  ``amplitudes`` are the
    controls themselves here: callables of t or arrays on the intervals."""

    def __init__(self, ops, amplitudes):
        self.ops = list(ops)
        self.amplitudes = list(amplitudes)

    def evaluate(self, vals):
        """``evaluate(generator, tlist, n; vals_dict)`` --
        src/generators.jl:740-754, with the control values already looked up."""
        return Operator(self.ops, list(vals))


# --------------------------------------------------------------------------
# index work at the boundary (CSC -> CSR, row partition)
# --------------------------------------------------------------------------

def csc_to_csr(n_rows, n_cols, colptr, rowval, nzval, index_base=1):
    """Convert Julia ``SparseMatrixCSC{ComplexF64,Int64}`` fields (1-based by
    default) to 0-based CSR with int64 rowptr / int32 col, columns ascending
    within each row (stable counting sort, the classic CSC->CSR transpose)."""
    colptr = np.asarray(colptr, dtype=np.int64) - index_base
    rowval = np.asarray(rowval, dtype=np.int64) - index_base
    nnz = int(colptr[-1])
    counts = np.bincount(rowval[:nnz], minlength=n_rows)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    col_of = np.repeat(np.arange(n_cols, dtype=np.int64), np.diff(colptr))
    order = np.argsort(rowval[:nnz], kind="stable")
    col = col_of[order].astype(np.int32)
    vals = np.asarray(nzval)[:nnz][order]
    return rowptr, col, vals


def partition_rows(rowptr, n_parts, balance="rows"):
    """Contiguous row blocks for G ranks.  ``rows``: equal row
    counts (first N % G ranks get one extra).  ``nnz``: boundaries at the
    first row whose prefix nnz reaches k * nnz / G."""
    n = len(rowptr) - 1
    if balance == "rows":
        base, rem = divmod(n, n_parts)
        bounds = [0]
        for r in range(n_parts):
            bounds.append(bounds[-1] + base + (1 if r < rem else 0))
        return np.array(bounds, dtype=np.int64)
    nnz = int(rowptr[-1])
    bounds = [0]
    for k in range(1, n_parts):
        target = (k * nnz) // n_parts
        bounds.append(int(np.searchsorted(rowptr, target, side="left")))
    bounds.append(n)
    return np.array(bounds, dtype=np.int64)


# --------------------------------------------------------------------------
# PWC propagator glue (src/propagator.jl, src/pwc_utils.jl,
# src/cheby_propagator.jl, src/newton_propagator.jl, src/propagate.jl)
# --------------------------------------------------------------------------

def get_tlist_midpoints(tlist):
    """src/controls.jl:92-124 (preserve_start = preserve_end = true)."""
    tlist = np.asarray(tlist, dtype=np.float64)
    N = len(tlist)
    assert N >= 3
    mid = np.zeros(N - 1)
    mid[0] = tlist[0]
    mid[-1] = tlist[-1]
    for i in range(1, N - 2):
        dt = tlist[i + 1] - tlist[i]
        mid[i] = tlist[i] + 0.5 * dt
    return mid


def discretize_on_midpoints(control, tlist):
    """src/controls.jl:189-208."""
    if callable(control):
        return np.array([float(control(t)) for t in get_tlist_midpoints(tlist)])
    control = np.asarray(control, dtype=np.float64)
    if len(control) == len(tlist) - 1:
        return control.copy()
    if len(control) == len(tlist):
        vals = np.empty(len(tlist) - 1)
        vals[0] = control[0]
        vals[-1] = control[-1]
        for i in range(1, len(vals) - 1):
            vals[i] = 2 * control[i] - vals[i - 1]
        return vals
    raise ValueError("control array must be defined on the points of tlist")


def discretize(control, tlist):
    """src/controls.jl:43-68."""
    if callable(control):
        return discretize(discretize_on_midpoints(control, tlist), tlist)
    control = np.asarray(control, dtype=np.float64)
    if len(control) == len(tlist):
        return control.copy()
    if len(control) == len(tlist) - 1:
        vals = np.zeros(len(control) + 1)
        vals[0] = control[0]
        vals[-1] = control[-1]
        for i in range(1, len(vals) - 1):
            vals[i] = 0.5 * (control[i - 1] + control[i])
        return vals
    raise ValueError("control array must be defined on intervals of tlist")


def get_uniform_dt(tlist, tol=1e-12):
    """src/propagator.jl:267-280."""
    dt = float(tlist[1] - tlist[0])
    for i in range(1, len(tlist) - 1):
        if abs((tlist[i + 1] - tlist[i]) - dt) > tol:
            return None
    return dt


class PWCPropagator:
    pass


def _genop(generator, vals):
    if isinstance(generator, Generator):
        return generator.evaluate(vals)
    return generator  # a static operator is its own genop (controls.jl:309-313)


def init_prop(state, generator, tlist, method, backward=False, parameters=None,
              control_ranges=None, specrange_method="auto", specrange_buffer=0.01,
              cheby_coeffs_limit=1e-12, check_normalization=False,
              uniform_dt_tolerance=1e-12, m_max=10, func=None, norm_min=1e-14,
              relerr=1e-12, max_restarts=50, specrange_state=None, **specrange_kwargs):
    """src/cheby_propagator.jl:87-175 and src/newton_propagator.jl:62-113."""
    if isinstance(generator, tuple) and len(generator) == 1 and not isinstance(generator[0], (tuple, list)):
        # `(H,)`: a tuple generator without controls evaluates to its drift matrix
        # (src/controls.jl:442-463; test/test_propagate.jl:22,157)
        generator = generator[0]
    p = PWCPropagator()
    p.method = method.lower()
    p.generator = generator
    p.tlist = np.asarray(tlist, dtype=np.float64)
    p.backward = backward
    p.inplace = True
    p.state = np.array(state, dtype=np.complex128)  # copy when in-place (:158)
    controls = generator.amplitudes if isinstance(generator, Generator) else []
    p.controls = controls
    if parameters is None:
        parameters = [discretize_on_midpoints(c, p.tlist) for c in controls]
    p.parameters = parameters
    p.n = 1
    p.t = float(p.tlist[0])
    if backward:
        p.n = len(p.tlist) - 1
        p.t = float(p.tlist[p.n])
    if p.method == "cheby":
        controlvals = [discretize(c, p.tlist) for c in controls]
        if control_ranges is None:
            control_ranges = [(float(np.min(v)), float(np.max(v))) for v in controlvals]
        p.control_ranges = control_ranges
        G_min = _genop(generator, [r[0] for r in control_ranges])
        G_max = _genop(generator, [r[1] for r in control_ranges])
        E_min, E_max = specrange(G_max, specrange_method, state=specrange_state,
                                 **specrange_kwargs)           # :340
        _E_min, _E_max = specrange(G_min, specrange_method, state=specrange_state,
                                   **specrange_kwargs)         # :341
        E_min = min(_E_min, E_min)
        E_max = max(_E_max, E_max)
        Delta = E_max - E_min
        assert Delta > 0.0
        delta = specrange_buffer * Delta                        # :131-133
        E_min = E_min - delta / 2
        Delta = Delta + delta
        dt = get_uniform_dt(p.tlist, tol=uniform_dt_tolerance)
        if dt is None:
            raise RuntimeError("Chebychev propagation only works on a uniform time grid")
        p.wrk = ChebyWrk(p.state, Delta, E_min, dt, limit=cheby_coeffs_limit)
        p.check_normalization = check_normalization
    elif p.method == "newton":
        p.wrk = NewtonWrk(p.state, m_max=m_max)
        p.func, p.norm_min, p.relerr, p.max_restarts = func, norm_min, relerr, max_restarts
    else:
        raise ValueError(f"Unknown propagation `method`: {method}")
    return p


def prop_step(p):
    """src/cheby_propagator.jl:348-386 / src/newton_propagator.jl:120-153 and
    src/pwc_utils.jl:102-112.  ``p.n`` keeps Julia's 1-based interval index."""
    n = p.n
    if not (0 < n < len(p.tlist)):
        return None
    vals = [par[n - 1] for par in p.parameters]
    H = _genop(p.generator, vals)
    if p.method == "cheby":
        dt = -p.wrk.dt if p.backward else p.wrk.dt
        cheby(p.state, H, dt, p.wrk, check_normalization=p.check_normalization)
    else:
        dt = p.tlist[n] - p.tlist[n - 1]
        if p.backward:
            dt = -dt
        newton(p.state, H, dt, p.wrk, func=p.func, norm_min=p.norm_min,
               relerr=p.relerr, max_restarts=p.max_restarts)
    if p.backward:                                              # pwc_utils.jl:102-112
        p.t = float(p.tlist[n - 1])
        p.n = n - 1
    else:
        p.t = float(p.tlist[n])
        p.n = n + 1
    return p.state


def propagate(state, generator, tlist, method, storage=False, **kw):
    """src/propagate.jl:283-344 (no observables/callback; optional storage of
    every state, column i = state at tlist[i])."""
    p = init_prop(state, generator, tlist, method, **kw)
    nt = len(p.tlist)
    store = None
    if storage:
        store = np.zeros((len(p.state), nt), dtype=np.complex128)
        store[:, (nt - 1) if p.backward else 0] = p.state
    for i in range(nt - 1):
        prop_step(p)
        if storage:
            store[:, (nt - 2 - i) if p.backward else (i + 1)] = p.state
    return (p.state, store) if storage else p.state
