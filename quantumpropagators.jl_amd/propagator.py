"""Host-side mirror of the reference's propagator interface for the two hot-path methods.

    init_prop(state, generator, tlist; method=Cheby|Newton, ...)   src/propagator.jl:208-264,
                                       src/cheby_propagator.jl:87-175, src/newton_propagator.jl:62-113
    prop_step!(propagator)             src/cheby_propagator.jl:348-386, src/newton_propagator.jl:120-153
    reinit_prop!(propagator, state)    src/cheby_propagator.jl:243-299, src/propagator.jl:298-312
    set_state!(propagator, state)      src/propagator.jl:367-377
    set_t!(propagator, t)              src/pwc_utils.jl:48-71
    propagate(state, generator, tlist) src/propagate.jl:167-344

Julia's ``!`` cannot appear in Python identifiers: ``prop_step!`` is ``prop_step`` etc.
Same argument names, meaning and error behaviour; the numerical work is done by
``libqprop_hip.so`` on the GPU (no CPU fallback).  ``propagator.n`` keeps Julia's 1-based
interval index so that the bookkeeping of src/pwc_utils.jl is restated verbatim.
"""
from __future__ import annotations

import bisect
import math
import warnings

import numpy as np
import scipy.sparse as sp

from . import lib as L

__all__ = ["Generator", "hamiltonian", "liouvillian", "MatrixFreeLiouvillian", "PauliSum", "init_prop", "prop_step", "reinit_prop", "set_state", "set_t",
           "propagate", "ode_function", "QuantumODEFunction", "ChebyPropagator", "NewtonPropagator", "discretize", "discretize_on_midpoints"]


# ----------------------------------------------------------------------------------------
# controls (src/controls.jl) -- scalar host logic, O(#steps)
# ----------------------------------------------------------------------------------------

def get_tlist_midpoints(tlist, preserve_start=True, preserve_end=True):
    """src/controls.jl:92-124: midpoints of the (not necessarily uniform) intervals; by default the
    first and the last point of ``tlist`` are kept instead of the outer midpoints."""
    tlist = np.asarray(tlist, dtype=np.float64)
    N = len(tlist)
    if N < 3:
        raise ValueError("In `get_tlist_midpoints`, argument `tlist` must have a length of at least 3")
    if np.any(np.diff(tlist) <= 0.0):
        raise AssertionError("dt > 0.0")                                  # the `@assert dt > 0.0` of :107-120
    mid = tlist[:-1] + 0.5 * np.diff(tlist)
    if preserve_start:
        mid[0] = tlist[0]
    if preserve_end:
        mid[-1] = tlist[-1]
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
        vals[0], vals[-1] = control[0], control[-1]
        for i in range(1, len(vals) - 1):
            vals[i] = 2 * control[i] - vals[i - 1]
        return vals
    raise ValueError("control array must be defined on the points of tlist")


def discretize(control, tlist, via_midpoints=True):
    """src/controls.jl:43-68."""
    if callable(control):
        if via_midpoints:
            return discretize(discretize_on_midpoints(control, tlist), tlist)
        return np.array([float(control(t)) for t in tlist])
    control = np.asarray(control, dtype=np.float64)
    if len(control) == len(tlist):
        return control.copy()
    if len(control) == len(tlist) - 1:
        vals = np.zeros(len(control) + 1)
        vals[0], vals[-1] = control[0], control[-1]
        for i in range(1, len(vals) - 1):
            vals[i] = 0.5 * (control[i - 1] + control[i])
        return vals
    raise ValueError("control array must be defined on intervals of tlist")


def _get_uniform_dt(tlist, tol=1e-12, warn=False):
    """src/propagator.jl:267-280."""
    dt = float(tlist[1] - tlist[0])
    for i in range(1, len(tlist) - 1):
        dt_i = tlist[i + 1] - tlist[i]
        if abs(dt_i - dt) > tol:
            if warn:
                warnings.warn(f"Non-uniform time grid: dt = {dt_i:.2e} in interval {i + 1} differs from the "
                              f"first dt={dt:.2e} by {abs(dt_i - dt):.2e} > tol = {tol:.2e}")
            return None
    return dt


# ----------------------------------------------------------------------------------------
# generators (src/generators.jl:44-61, :388-469)
# ----------------------------------------------------------------------------------------

class Generator:
    """sum_l a_l(t) H_l: ``ops`` (matrices), ``amplitudes`` for the last len(amplitudes)
    ops (drift terms first, src/generators.jl:44-61).  Amplitudes here are the controls
    themselves: callables of t, or arrays on tlist / on the intervals."""

    def __init__(self, ops, amplitudes):
        self.ops = list(ops)
        self.amplitudes = list(amplitudes)
        if len(self.amplitudes) > len(self.ops):
            raise ValueError("more amplitudes than operators")


def hamiltonian(*terms):
    """``hamiltonian(H0, (H1, eps1), ...)`` -- src/generators.jl:388-469 (``_make_generator``): the
    drift terms are summed into one operator listed first; terms that share an amplitude (the same
    function object, or equal arrays) are merged by summing their operators; a term must be an
    operator or a 2-tuple ``(op, amplitude)``.  Without time-dependent terms the drift operator
    itself is returned."""
    drift = None
    ops, ampl = [], []
    for t in terms:
        if isinstance(t, (tuple, list)):
            if len(t) != 2:
                raise ValueError("time-dependent term must be 2-tuple")                   # :401
            op, a = t
            k = next((i for i, b in enumerate(ampl)
                      if b is a or (isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and np.array_equal(a, b))),
                     None)
            if k is None:
                ops.append(op)
                ampl.append(a)
            else:
                ops[k] = ops[k] + op                                                      # :412-424
        else:
            drift = t if drift is None else drift + t
    if drift is not None:
        ops.insert(0, drift)
    if not ampl:
        if not ops:
            raise ValueError("Generator has no terms")                                    # :452
        return ops[0]
    return Generator(ops, ampl)


class PauliSum:
    """A qubit-register operator given as a sum of Pauli strings, ``PauliSum(n, [(amplitude, 'XIZY' | {qubit: 'X'} | (xmask, zmask)), ...])``
    -- usable wherever :func:`hamiltonian` takes a matrix: ``hamiltonian(PauliSum(n, zz), (PauliSum(n, x), eps))``.  A generator
    whose operators are all PauliSums goes to the device as ``lib.PauliOperator`` (applied from the bit masks: no stored matrix,
    include/qprop.h qp_pauli_operator_create); ``+`` concatenates the strings, ``toarray()`` / ``tocsr()`` build the matrix (what the
    reference would be handed: the oracle's input in the tests)."""

    def __init__(self, nqubits, strings):
        self.nqubits = int(nqubits)
        self.strings = [(complex(a), st if (isinstance(st, tuple) and len(st) == 2 and not isinstance(st[0], str)) else L.pauli_masks(st))
                        for a, st in strings]
        self.shape = (1 << self.nqubits, 1 << self.nqubits)

    def __add__(self, other):
        if not isinstance(other, PauliSum) or other.nqubits != self.nqubits:
            return NotImplemented
        return PauliSum(self.nqubits, self.strings + other.strings)

    def tocsr(self):
        from . import synth
        return synth.pauli_sum_matrix(self.nqubits, self.strings)

    def toarray(self):
        return self.tocsr().toarray()


class MatrixFreeLiouvillian:
    """What :func:`liouvillian` returns for ``matrix_free=True``: the Hamiltonian terms, the
    amplitudes of the controlled ones and the Lindblad operators, kept as n x n matrices.  On
    the device it becomes ``lib.Liouvillian`` (GEMMs on rho) instead of an n^2 x n^2 sparse
    matrix."""

    def __init__(self, ops, amplitudes, c_ops, convention):
        self.ops, self.amplitudes, self.c_ops, self.convention = list(ops), list(amplitudes), list(c_ops), convention


def liouvillian(H, c_ops=(), *, convention, matrix_free=False):
    """``liouvillian(H, c_ops; convention)`` -- src/generators.jl:515-631.  ``H``: a matrix, a
    tuple ``(H0, (H1, eps1), ...)``, a :class:`Generator` or ``None``; ``convention`` is
    mandatory ("TDSE" or "LvN").  Returns the superoperator as sparse matrices (a
    :class:`Generator` with the dissipator folded into the drift when H is time-dependent), or,
    with ``matrix_free=True``, a :class:`MatrixFreeLiouvillian` for dense operators."""
    conv = str(convention).lstrip(":")
    if conv.upper() not in ("TDSE", "LVN"):
        raise ValueError("convention must be :TDSE or :LvN")                          # :486-488
    conv = "TDSE" if conv.upper() == "TDSE" else "LvN"
    if isinstance(H, (tuple, list)):
        H = hamiltonian(*H)
    if H is None and len(c_ops) == 0:
        raise ValueError("Empty Liouvillian, must give at least one of `H` or `c_ops`")  # :627-630
    ops = [] if H is None else (H.ops if isinstance(H, Generator) else [H])
    ampl = H.amplitudes if isinstance(H, Generator) else []
    if matrix_free:
        return MatrixFreeLiouvillian(ops, ampl, c_ops, conv)
    from . import synth
    drift_n = len(ops) - len(ampl)
    drift = None
    for A in c_ops:                                                                      # dissipator :512-520
        D = synth.lindblad_to_superop(A, conv)
        drift = D if drift is None else drift + D
    terms = []
    for i, op in enumerate(ops):
        Lh = synth.ham_to_superop(sp.csr_matrix(op), conv)
        if i < drift_n:
            drift = Lh if drift is None else drift + Lh
        else:
            terms.append((Lh, ampl[i - drift_n]))
    return hamiltonian(*([] if drift is None else [drift.tocsr()]), *terms)


def _to_matrix(ctx, A):
    if isinstance(A, L.Matrix):
        return A
    if sp.issparse(A):
        return L.Matrix.from_scipy(ctx, A)
    return L.Matrix.from_dense(ctx, np.asarray(A))


class _DeviceGenerator:
    """The generator's matrices uploaded once; `evaluate!` only rewrites coefficients
    (src/generators.jl:757-766) -- the only per-step host->device payload."""

    def __init__(self, ctx, generator, fmt=L.FMT_AUTO):
        if isinstance(generator, tuple) and len(generator) == 1:
            generator = generator[0]          # `(H,)` in test/test_propagate.jl:157
        if isinstance(generator, (tuple, list)):
            generator = hamiltonian(*generator)
        if isinstance(generator, MatrixFreeLiouvillian):
            self.controls = list(generator.amplitudes)
            self.op = L.Liouvillian(ctx, generator.ops, generator.c_ops, ncoeffs=len(self.controls),
                                    convention=generator.convention)
            return
        if isinstance(generator, PauliSum) or (isinstance(generator, Generator) and generator.ops and
                                               all(isinstance(A, PauliSum) for A in generator.ops)):
            ops = [generator] if isinstance(generator, PauliSum) else generator.ops
            self.controls = [] if isinstance(generator, PauliSum) else list(generator.amplitudes)
            self.op = L.PauliOperator(ctx, ops[0].nqubits, [A.strings for A in ops], ncoeffs=len(self.controls))
            return
        if isinstance(generator, Generator):
            self.controls = list(generator.amplitudes)
            mats = [_to_matrix(ctx, A.tocsr() if isinstance(A, PauliSum) else A) for A in generator.ops]
        elif isinstance(generator, L.Operator):
            self.controls = []
            self.op = generator
            return
        else:
            self.controls = []
            mats = [_to_matrix(ctx, generator)]
        self.op = L.Operator(ctx, mats, len(self.controls), fmt)

    def set_vals(self, vals):
        if self.controls:
            self.op.set_coeffs(np.asarray(vals, dtype=np.complex128))
        return self.op


class QuantumODEFunction:
    """``f = ode_function(generator, tlist; c=-1im)`` -- src/ode_function.jl:54-98: the right-hand
    side d|Psi>/dt = c H(t) |Psi> for an ODE solver.  ``f(du, u, p, t)`` writes into the device
    state ``du`` and returns it; ``f(u, p, t)`` returns a new device state.  ``p`` (the solver's
    parameters, the reference's ``vals_dict``) maps a control -- by identity -- to the value that
    replaces it; every other control must be a callable of ``t``
    (``evaluate(control::Vector, t)`` is an error, src/controls.jl:388-396).  The matrices stay
    on the device; a call uploads the coefficients and launches one ``mul!(du, H, u, c, false)``."""

    def __init__(self, ctx, generator, c=-1j, device_format=L.FMT_AUTO):
        self.ctx = ctx
        self.generator = generator
        self._dgen = _DeviceGenerator(ctx, generator, device_format)
        self.c = complex(c)

    @property
    def operator(self):
        return self._dgen.op

    def _evaluate(self, p, t):
        vals = []
        for control in self._dgen.controls:
            hit = None
            if p:
                items = p.items() if hasattr(p, "items") else p
                hit = next((v for k, v in items if k is control), None)
            if hit is not None:
                vals.append(hit)
            elif callable(control):
                vals.append(control(float(t)))
            else:
                raise TypeError("`evaluate(control::Vector, t::Float64)` is invalid. Use e.g. `evaluate(…, tlist, n)`.")
        return self._dgen.set_vals(vals)

    def __call__(self, *args):
        if len(args) == 4:
            du, u, p, t = args
        elif len(args) == 3:
            u, p, t = args
            du = L.State(self.ctx, n=u.n)
        else:
            raise TypeError("f(du, u, p, t) or f(u, p, t)")
        H = self._evaluate(p, t)
        return H.mul(u, du, self.c, 0.0)


def ode_function(generator, tlist=None, *, c=-1j, ctx=None, device=0, device_format=L.FMT_AUTO):
    """src/ode_function.jl:54-63.  ``tlist`` is accepted for signature parity (the reference uses
    it only to build the first operator)."""
    ctx = ctx if ctx is not None else L.Context(device)
    return QuantumODEFunction(ctx, generator, c=c, device_format=device_format)



# ----------------------------------------------------------------------------------------
# propagators
# ----------------------------------------------------------------------------------------

class PWCPropagator:
    """Common fields: state, tlist, t, n, parameters, backward, inplace
    (src/propagator.jl:119-126, src/pwc_utils.jl:5-24).  Property access follows
    src/propagator.jl:77-127: ``generator`` is hidden, ``state`` / ``t`` change only through
    ``set_state`` / ``set_t``, ``tlist`` is read-only, ``parameters`` may be re-bound."""

    PUBLIC_PROPERTIES = ("state", "tlist", "t", "parameters", "backward", "inplace")     # propertynames, :119-126
    _MESSAGES = {"state": "The state of a propagator can only be set via `set_state!`",
                 "tlist": "The tlist of a propagator is read-only",
                 "t": "The current time of a propagator can only be set via `set_t!`"}

    @property
    def generator(self):  # hidden, src/propagator.jl:77-86
        raise AttributeError(f"type {type(self).__name__} has no property generator")

    def __setattr__(self, name, value):
        if name == "generator":
            raise AttributeError(f"type {type(self).__name__} has no property generator")
        if name in self._MESSAGES and self.__dict__.get("_sealed", False):
            raise AttributeError(self._MESSAGES[name])
        object.__setattr__(self, name, value)

    def _set(self, name, value):
        """Internal write access for prop_step! / set_state! / set_t!."""
        object.__setattr__(self, name, value)

    def propertynames(self, private=False):
        if not private:
            return self.PUBLIC_PROPERTIES
        return tuple(dict.fromkeys(self.PUBLIC_PROPERTIES + tuple(self.__dict__)))


class ChebyPropagator(PWCPropagator):
    pass


class NewtonPropagator(PWCPropagator):
    pass


def _as_state(ctx, state, copy):
    if isinstance(state, L.State):
        if not copy:
            return state
        s = L.State(ctx, n=state.n)
        s.copy_from(state)
        return s
    return L.State(ctx, data=np.asarray(state, dtype=np.complex128))


def _specrange(op, method, ctx, n, rng=None, state=None, **kw):
    """``specrange(H, method; kw...)`` -- src/specrad.jl:36-140."""
    method = str(method).lower()
    if method == "auto":                                           # :45-61
        if "E_min" in kw and "E_max" in kw:
            method = "manual"
        elif n <= 32 and op.format != L.FMT_MATFREE:
            method = "diag"
        else:
            method = "arnoldi"
    if method == "manual":                                         # :138-140
        if "E_min" not in kw or "E_max" not in kw:
            raise TypeError("UndefKeywordError: keyword argument E_min/E_max not assigned")
        return float(kw["E_min"]), float(kw["E_max"])
    if method == "diag":                                           # :124-128 (tiny: host eigvals of Array(H))
        rp, col, vals = op.get_csr()
        A = sp.csr_matrix((vals, col, rp), shape=op.shape).toarray()
        ev = np.linalg.eigvals(A)
        ev = ev[np.lexsort((ev.imag, ev.real))].real
        return float(ev[0]), float(ev[-1])
    if method == "arnoldi":                                        # :88-112
        if state is None:                                          # random_state, :153-158
            rng = np.random.default_rng() if rng is None else rng
            psi = rng.random(n) * np.exp(2j * np.pi * rng.random(n))
            psi /= np.linalg.norm(psi)
            state = L.State(ctx, data=psi)
        elif not isinstance(state, L.State):
            state = L.State(ctx, data=state)
        return L.specrange_arnoldi(op, state, m_min=kw.get("m_min", 25), m_max=kw.get("m_max", 60),
                                   prec=kw.get("prec", 1e-3), norm_min=kw.get("norm_min", 1e-15),
                                   enlarge=kw.get("enlarge", True))
    raise ValueError(f"unknown specrange method {method!r}")


def _cheby_get_spectral_envelope(p, control_ranges, method, **kw):
    """src/cheby_propagator.jl:331-345."""
    n = p.state.n
    G_max = p._dgen.set_vals([r[1] for r in control_ranges])
    E_min, E_max = _specrange(G_max, method, p.ctx, n, **kw)
    G_min = p._dgen.set_vals([r[0] for r in control_ranges])
    _E_min, _E_max = _specrange(G_min, method, p.ctx, n, **kw)
    return min(_E_min, E_min), max(_E_max, E_max)


def _method_name(method):
    name = getattr(method, "__name__", method)
    name = str(name).split(".")[-1].lower()
    if name not in ("cheby", "newton"):
        raise ValueError(f"Unknown propagation `method`: {method}")      # src/propagator.jl:262-264
    return name


def init_prop(state, generator, tlist, method, *, inplace=True, backward=False, verbose=False,
              parameters=None, piecewise=None, pwc=None, ctx=None, device=0, device_format=L.FMT_AUTO,
              # Cheby
              control_ranges=None, specrange_method="auto", specrange_buffer=0.01,
              cheby_coeffs_limit=1e-12, check_normalization=False, uniform_dt_tolerance=1e-12,
              # Newton
              m_max=10, func=None, norm_min=1e-14, relerr=1e-12, max_restarts=50,
              **specrange_kwargs):
    name = _method_name(method)
    ctx = ctx if ctx is not None else (state.ctx if isinstance(state, L.State) else L.Context(device))
    tlist = np.asarray(tlist, dtype=np.float64)
    p = ChebyPropagator() if name == "cheby" else NewtonPropagator()
    p.ctx = ctx
    p.method = name
    p.tlist = tlist
    p.backward = bool(backward)
    p.inplace = bool(inplace)
    p._dgen = _DeviceGenerator(ctx, generator, device_format)
    p.controls = p._dgen.controls
    if name == "newton" and not inplace:
        raise RuntimeError("The Newton propagator is only implemented in-place")   # newton_propagator.jl:94
    if parameters is None:                                              # pwc_utils.jl:29-45
        parameters = [discretize_on_midpoints(c, tlist) for c in p.controls]
    else:
        parameters = list(parameters)
        for amp in parameters:
            assert len(amp) == len(tlist) - 1
    p.parameters = parameters
    p.state = _as_state(ctx, state, copy=inplace)                       # copy when in-place (:158)
    p.n = 1
    p.t = float(tlist[0])
    if backward:
        p.n = len(tlist) - 1
        p.t = float(tlist[p.n])
    if name == "cheby":
        controlvals = [discretize(c, tlist) for c in p.controls]
        if control_ranges is None:
            control_ranges = [(float(np.min(v)), float(np.max(v))) for v in controlvals]
        else:
            control_ranges = [tuple(r) for r in control_ranges]
            for r in control_ranges:
                assert r[0] <= r[1]
        p.control_ranges = control_ranges
        p.specrange_method = specrange_method
        p.specrange_buffer = float(specrange_buffer)
        p.specrange_options = dict(specrange_kwargs)
        p.check_normalization = bool(check_normalization)
        E_min, E_max = _cheby_get_spectral_envelope(p, control_ranges, specrange_method, **specrange_kwargs)
        Delta = E_max - E_min
        assert Delta > 0.0
        delta = specrange_buffer * Delta                                # :131-133
        E_min = E_min - delta / 2
        Delta = Delta + delta
        dt = _get_uniform_dt(tlist, tol=uniform_dt_tolerance, warn=True)
        if dt is None:
            raise RuntimeError("Chebychev propagation only works on a uniform time grid")
        p.wrk = L.ChebyWrk(ctx, p.state.n, Delta, E_min, dt, limit=cheby_coeffs_limit)
    else:
        p.wrk = L.NewtonWrk(ctx, p.state.n, m_max=m_max)
        p.func, p.norm_min, p.relerr, p.max_restarts = func, norm_min, relerr, max_restarts
    if piecewise is True or pwc is True:
        pass  # both propagators are piecewise-constant (src/propagator.jl:232-245)
    p._set("_sealed", True)
    return p


def prop_step(p):
    """``prop_step!(propagator)``: returns ``propagator.state`` (the same object when
    in-place) or ``None`` past the end of the grid."""
    n = p.n
    tlist = p.tlist
    if not (0 < n < len(tlist)):
        return None
    H = p._dgen.set_vals([par[n - 1] for par in p.parameters])          # _pwc_set_genop!  pwc_utils.jl:86-92
    if not p.inplace:
        new = L.State(p.ctx, n=p.state.n)
        new.copy_from(p.state)
        p._set("state", new)
    if p.method == "cheby":
        dt = -p.wrk.dt if p.backward else p.wrk.dt
        L.cheby(p.state, H, dt, p.wrk, check_normalization=p.check_normalization)
    else:
        dt = tlist[n] - tlist[n - 1]                                    # newton_propagator.jl:127-130
        if p.backward:
            dt = -dt
        L.newton(p.state, H, dt, p.wrk, func=p.func, norm_min=p.norm_min, relerr=p.relerr,
                 max_restarts=p.max_restarts)
    if p.backward:                                                      # _pwc_advance_time!  pwc_utils.jl:102-112
        p._set("t", float(tlist[n - 1]))
        p.n = n - 1
    else:
        p._set("t", float(tlist[n]))
        p.n = n + 1
    return p.state


def set_state(p, state):
    """src/propagator.jl:367-377."""
    if state is not p.state:
        if p.inplace:
            if isinstance(state, L.State):
                p.state.copy_from(state)
            else:
                p.state.upload(np.asarray(state, dtype=np.complex128))
        else:
            p._set("state", _as_state(p.ctx, state, copy=False))
    return p.state


def set_t(p, t):
    """``set_t!`` = ``_pwc_set_t!`` -- src/pwc_utils.jl:48-71 (snaps with a warning)."""
    tlist = p.tlist
    if t <= tlist[0]:
        n = 1
    else:
        N = len(tlist)
        if t >= tlist[-1]:
            n = N
        else:
            n = min(bisect.bisect_left(tlist.tolist(), t) + 1, N)
    if not math.isclose(t, tlist[n - 1], rel_tol=math.sqrt(np.finfo(float).eps)):
        warnings.warn(f"Snapping t={t} to time grid value {tlist[n - 1]}")
    p.n = n - 1 if p.backward else n
    p._set("t", float(tlist[n - 1]))


def reinit_prop(p, state, transform_control_ranges=None, **_):
    """src/cheby_propagator.jl:243-299 (Cheby: coefficients are recalculated when the
    current ``parameters`` exceed the stored control ranges) / src/propagator.jl:298-312."""
    state = set_state(p, state)
    if p.method == "cheby":
        tcr = transform_control_ranges or (lambda c, lo, hi, check: (lo, hi))
        ranges = [(float(np.min(par)), float(np.max(par))) for par in p.parameters]
        need = False
        for c, (lo, hi), (slo, shi) in zip(p.controls, ranges, p.control_ranges):
            lo_c, hi_c = tcr(c, lo, hi, True)
            if lo_c < slo or hi_c > shi:
                need = True
                break
        if need:
            ranges = [tuple(tcr(c, lo, hi, False)) for c, (lo, hi) in zip(p.controls, ranges)]
            E_min, E_max = _cheby_get_spectral_envelope(p, ranges, p.specrange_method, **p.specrange_options)
            Delta = E_max - E_min
            assert Delta > 0.0
            delta = p.specrange_buffer * Delta
            E_min = E_min - delta / 2
            Delta = Delta + delta
            dt = float(p.tlist[1] - p.tlist[0])
            p.control_ranges = ranges
            p.wrk = L.ChebyWrk(p.ctx, state.n, Delta, E_min, dt, limit=p.wrk.limit)
    set_t(p, float(p.tlist[-1] if p.backward else p.tlist[0]))


def _is_matrix_observable(o):
    return sp.issparse(o) or (isinstance(o, np.ndarray) and o.ndim == 2)


def _propagate_fused(p, storage, observables):
    """The whole step loop in one ``qp_propagate`` call (no per-step host round trip):
    possible when nothing has to call back into the host between steps."""
    nt = len(p.tlist)
    nsteps = nt - 1
    order = range(nsteps, 0, -1) if p.backward else range(1, nt)        # interval index n per step
    if p.method == "cheby":
        dts = np.full(nsteps, -p.wrk.dt if p.backward else p.wrk.dt)
    else:
        dts = np.array([p.tlist[n] - p.tlist[n - 1] for n in order])
        if p.backward:
            dts = -dts
    table = None
    if p.controls:
        table = np.array([[par[n - 1] for par in p.parameters] for n in order], dtype=np.complex128)
    obs_ops = []
    if storage and observables is not None:
        obs_ops = [L.Operator(p.ctx, [_to_matrix(p.ctx, o)], 0, L.FMT_AUTO) for o in observables]
    kw = {}
    if p.method == "newton":
        kw = dict(func=p.func, norm_min=p.norm_min, relerr=p.relerr, max_restarts=p.max_restarts)
    else:
        kw = dict(check_normalization=p.check_normalization)
    H = p._dgen.op
    ev, st = L.propagate_steps(H, p.state, p.wrk, dts, table, observables=obs_ops,
                               store_states=bool(storage) and observables is None, **kw)
    p.n = 0 if p.backward else nt
    p._set("t", float(p.tlist[0] if p.backward else p.tlist[-1]))
    rows = ev if obs_ops else st
    if rows is None:
        return None
    store = rows.T.copy()                                               # column i <-> tlist[i]
    return store[:, ::-1].copy() if p.backward else store


def propagate(state, generator, tlist, *, method, backward=False, inplace=True, storage=None,
              observables=None, callback=None, fused=None, **kwargs):
    """``propagate(state, generator, tlist; method, ...)`` -- src/propagate.jl:167-344.
    ``storage=True`` returns (final_state, array) with column i = state at tlist[i]
    (or the observables' values).  States come back as NumPy arrays.

    Observables are callables ``o(psi)`` or matrices (expectation value ``dot(psi, O, psi)``,
    src/storage.jl:121-123).  Without a callback and with matrix observables (or none) the step
    loop runs inside the library (``fused``, default on in that case)."""
    p = init_prop(state, generator, tlist, method, backward=backward, inplace=inplace, **kwargs)
    nt = len(p.tlist)
    can_fuse = (callback is None and p.inplace and
                (observables is None or all(_is_matrix_observable(o) for o in observables)))
    if fused is None:
        fused = can_fuse
    elif fused and not can_fuse:
        raise ValueError("fused propagation needs inplace=True, no callback and matrix observables")
    if fused:
        store = _propagate_fused(p, storage, observables)
        out = p.state.numpy()
        return (out, store) if storage else out

    import inspect

    def n_positional(f):
        try:
            return sum(1 for q in inspect.signature(f).parameters.values()
                       if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD) and q.default is q.empty)
        except (TypeError, ValueError):
            return 1
    arity = [0 if _is_matrix_observable(o) else n_positional(o) for o in (observables or ())]

    def one(k, o, psi, slot):
        """map_observable (src/storage.jl:100-123): a matrix -> dot(psi, O, psi); a function of
        ``(state, tlist, i)`` or of ``state`` (``i`` is the 0-based index into tlist here)."""
        if arity[k] == 0:
            return np.vdot(psi, o @ psi)
        if arity[k] >= 3:
            return o(psi, p.tlist, slot)
        return o(psi)

    def obs(s, slot):
        psi = s.numpy()
        if observables is None:
            return psi
        return np.array([one(k, o, psi, slot) for k, o in enumerate(observables)])
    store = None
    if storage:
        slot0 = (nt - 1) if backward else 0
        first = obs(p.state, slot0)
        store = np.zeros((len(first), nt), dtype=first.dtype)
        store[:, slot0] = first
    for i in range(nt - 1):
        prop_step(p)
        if callback is not None:
            callback(p, observables)
        if storage:
            slot = (nt - 2 - i) if backward else (i + 1)
            store[:, slot] = obs(p.state, slot)
    out = p.state.numpy()
    return (out, store) if storage else out
