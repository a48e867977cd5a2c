"""``QuantumPropagators.Interfaces`` for the device types: the acceptance checks a user of the
reference runs on custom state / operator / propagator types, restated for ``lib.State``,
``lib.Operator`` and the propagators of ``propagator.py``.

* :func:`check_state`      -- src/interfaces/state.jl:92-400 (immutable + in-place interface)
* :func:`check_operator`   -- src/interfaces/operator.jl:72-260 (size, ``op * state``, 3- and
  5-argument ``mul!``, ``dot(state, op, state)``)
* :func:`check_propagator` -- src/interfaces/propagator.jl:55-338

Each returns ``True`` / ``False`` and, unless ``quiet``, reports every violated requirement
through :mod:`warnings` (the reference logs them with ``@error``)."""
import warnings

import numpy as np

from . import lib as L
from . import propagator as P


class _Report:
    def __init__(self, quiet, prefix):
        self.ok, self.quiet, self.prefix = True, quiet, prefix

    def fail(self, msg):
        self.ok = False
        if not self.quiet:
            warnings.warn(f"{self.prefix}{msg}")

    def attempt(self, what, fn):
        try:
            return fn()
        except Exception as exc:          # noqa: BLE001 -- mirrors the reference's catch-all
            self.fail(f"{what} must be defined ({type(exc).__name__}: {exc})")
            return None


def check_state(state, *, for_immutable_state=True, for_mutable_state=True, normalized=False,
                atol=1e-14, quiet=False, _message_prefix=""):
    """src/interfaces/state.jl:92-400."""
    r = _Report(quiet, _message_prefix)
    if for_immutable_state:
        c = r.attempt("`state . state`", lambda: state.dot(state))
        if c is not None:
            if not isinstance(c, complex):
                r.fail(f"`state . state` must return a complex number, not {type(c)}")
            n1 = r.attempt("`norm(state)`", state.norm)
            if n1 is not None and abs(n1 - np.sqrt(abs(c))) > atol:                        # :124-139
                r.fail(f"`norm(state)={n1}` must match sqrt(state . state)={np.sqrt(abs(c))}")
        d = r.attempt("`state - state`", lambda: (state - state).norm())                    # :141-161
        if d is not None and d > atol:
            r.fail("`state - state` must have norm 0")
        s2 = r.attempt("`state + state`", lambda: (state + state).norm())
        if s2 is not None and s2 > 2 * state.norm() + atol:
            r.fail("`norm(state + state)` must fulfill the triangle inequality")
        cp = r.attempt("`copy(state)`", state.copy)                                         # :163-181
        if cp is not None:
            if type(cp) is not type(state):
                r.fail("`copy(state)` must have the same type as `state`")
            elif (cp - state).norm() > atol:
                r.fail("`copy(state) - state` must have norm 0")
        h = r.attempt("`a * state`", lambda: (0.5 * state).norm())                          # :183-197
        if h is not None and abs(h - 0.5 * state.norm()) > atol:
            r.fail("`norm(state)` must have absolute homogeneity: norm(s * state) = s * norm(state)")
        z = r.attempt("`zero(state)`", lambda: state.zero().norm())                         # :199-211
        if z is not None and z > atol:
            r.fail("`zero(state)` must produce a state with norm 0")
        z = r.attempt("`0.0 * state`", lambda: (0.0 * state).norm())                        # :213-225
        if z is not None and z > atol:
            r.fail("`0.0 * state` must produce a state with norm 0")
    if for_mutable_state:
        phi = r.attempt("`similar(state)`", state.similar)                                  # :227-282
        if phi is not None:
            if type(phi) is not type(state):
                r.fail("`similar(state)` must have the same type as `state`")
            else:
                got = r.attempt("`copyto!(other, state)`", lambda: phi.copy_from(state))
                if got is not None and (phi - state).norm() > atol:
                    r.fail("`phi - state` must have norm 0, where phi = similar(state); copyto!(phi, state)")
        phi = r.attempt("`copy(state)`", state.copy)
        if phi is not None:
            ret = r.attempt("`fill!(state, 0.0)`", lambda: phi.fill(0.0))                   # :284-305
            if ret is not None:
                if ret is not phi:
                    r.fail("`fill!(state, 0.0)` must return the filled state")
                if phi.norm() > atol:
                    r.fail("`fill!(state, 0.0)` must have norm 0")
            phi.copy_from(state)
            ret = r.attempt("`lmul!(c, state)`", lambda: phi.lmul(0.5))                     # :307-325
            if ret is not None and abs(phi.norm() - 0.5 * state.norm()) > atol:
                r.fail("`norm(state)` must have absolute homogeneity: norm(s * state) = s * norm(state)")
            ret = r.attempt("`lmul!(0.0, state)`", lambda: phi.lmul(0.0))                   # :327-345
            if ret is not None and phi.norm() > atol:
                r.fail("`lmul!(0.0, state)` must produce a state with norm 0")
            phi.copy_from(state)
            ret = r.attempt("`axpy!(c, state, other)`", lambda: phi.axpy(0.5, state))       # :347-365
            if ret is not None and (phi - 1.5 * state).norm() > atol:
                r.fail("`axpy!(a, state, phi)` must match `phi += a * state`")
    if normalized:                                                                          # :367-383
        eta = state.norm()
        if abs(eta - 1.0) > atol:
            r.fail(f"`norm(state)` must be 1, not {eta}")
    return r.ok


def check_operator(op, *, state, tlist=(0.0, 1.0), for_mutable_state=True, for_immutable_state=True,
                   atol=1e-14, quiet=False, _message_prefix=""):
    """src/interfaces/operator.jl:72-260 (the part that does not assume a host array)."""
    r = _Report(quiet, _message_prefix)
    if not check_state(state, for_mutable_state=for_mutable_state, for_immutable_state=for_immutable_state,
                       atol=atol, quiet=quiet, _message_prefix=_message_prefix + "On `state`: "):
        r.fail("The `state` is not a valid state")                                          # :95-99
        return False
    s = r.attempt("`size(op)`", op.size)                                                    # :101-131
    if s is not None:
        if not (isinstance(s, tuple) and all(isinstance(n, (int, np.integer)) for n in s)):
            r.fail(f"`size(op)` must return a tuple of integers, not {s}")
        else:
            for dim, n in enumerate(s):
                if op.size(dim) != n:
                    r.fail(f"`size(op, {dim})` must be consistent with `size(op)`")
    phi = None
    if for_immutable_state:
        phi = r.attempt("`op * state`", lambda: op * state)                                 # :162-176
        if phi is not None and type(phi) is not type(state):
            r.fail(f"`op * state` must return an object of the same type as `state`, not {type(phi)}")
    if for_mutable_state:
        out = state.similar()
        ret = r.attempt("`mul!(phi, op, state)`", lambda: op.mul(state, out))               # :178-197
        if ret is not None:
            if ret is not out:
                r.fail("`mul!(phi, op, state)` must return the resulting phi")
            if phi is not None and (out - phi).norm() > atol:
                r.fail("`mul!(phi, op, state)` must match `op * state`")
        out = state.copy()
        ret = r.attempt("`mul!(phi, op, state, alpha, beta)`", lambda: op.mul(state, out, 0.5, 0.5))   # :199-221
        if ret is not None:
            if ret is not out:
                r.fail("`mul!(phi, op, state, alpha, beta)` must return the resulting phi")
            if phi is not None and (out - (0.5 * state + 0.5 * phi)).norm() > atol:
                r.fail("`mul!(phi, op, state, alpha, beta)` must match beta*phi + alpha*op*state")
    val = r.attempt("`dot(state, op, state)`", lambda: op.dot(state, state))                # :223-256
    if val is not None:
        if not isinstance(val, complex):
            r.fail(f"`dot(state, op, state)` must return a number, not {type(val)}")
        elif phi is not None and abs(val - state.dot(phi)) > atol * max(1.0, abs(val)):
            r.fail("`dot(state, op, state)` must match `dot(state, op * state)`")
    return r.ok


def check_propagator(propagator, *, atol=1e-14, quiet=False, _message_prefix=""):
    """src/interfaces/propagator.jl:55-338: runs the propagator over its whole time grid and
    back to the start with ``reinit_prop!``."""
    r = _Report(quiet, _message_prefix)
    p = propagator
    tlist = p.tlist
    if not check_state(p.state, for_mutable_state=p.inplace, atol=1e-12, quiet=quiet,
                       _message_prefix=_message_prefix + "On `propagator.state`: "):
        r.fail("`propagator.state` is not a valid state")
    if p.t != (tlist[-1] if p.backward else tlist[0]):                                      # :111-127
        r.fail("`propagator.t` must be the initial (backward: final) value of `propagator.tlist`")
    for par in p.parameters:                                                                # :129-146
        if len(par) != len(tlist) - 1:
            r.fail("the values of `propagator.parameters` must have one element per interval of tlist")
    psi0 = p.state.copy()
    s0 = p.state
    s1 = r.attempt("`prop_step!(propagator)`", lambda: P.prop_step(p))                      # :148-215
    if s1 is not None:
        if p.inplace and s1 is not s0:
            r.fail("For an in-place propagator, the state returned by `prop_step!` must be the `propagator.state` object")
        if not p.inplace and s1 is s0:
            r.fail("For a not-in-place propagator, `prop_step!` must return a new object")
        if s1 is not p.state:
            r.fail("`prop_step!` must return `propagator.state`")
        if abs(p.t - (tlist[-2] if p.backward else tlist[1])) > atol:
            r.fail("`prop_step!` must advance `propagator.t` forward or backward one step on the time grid")
        psi1 = s1.copy()
        if r.attempt("`propagator.generator` must be hidden", lambda: _no_generator(p)) is False:
            r.fail("`propagator.generator` must not be accessible")
        while P.prop_step(p) is not None:                                                   # :217-262
            pass
        n_end, t_end = p.n, p.t
        if t_end != (tlist[0] if p.backward else tlist[-1]):
            r.fail("propagating to the end of the grid must leave `propagator.t` at its last point")
        if P.prop_step(p) is not None or p.n != n_end or p.t != t_end:
            r.fail("`prop_step!` must return `nothing` when going beyond the time grid, and leave the propagator unchanged")
        ret = r.attempt("`set_state!(propagator, state)`", lambda: P.set_state(p, psi0))   # :264-292
        if ret is not None and (ret is not p.state or (p.state - psi0).norm() > atol):
            r.fail("`set_state!` must return and set `propagator.state`")
        r.attempt("`set_t!(propagator, t)`", lambda: P.set_t(p, tlist[1]))                  # :294-306
        if p.t != tlist[1]:
            r.fail("`set_t!(propagator, t)` must set `propagator.t`")
        r.attempt("`reinit_prop!(propagator, state)`", lambda: P.reinit_prop(p, psi0))      # :308-335
        if p.t != (tlist[-1] if p.backward else tlist[0]):
            r.fail("`reinit_prop!` must reset `propagator.t`")
        again = P.prop_step(p)
        if again is None or (again - psi1).norm() > atol:
            r.fail("`reinit_prop!` must reset the propagator: the first step must be reproducible")
    return r.ok


def _no_generator(p):
    try:
        p.generator
    except AttributeError:
        return True
    return False
