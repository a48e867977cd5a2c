"""Pin the double-precision oracle (and the library's host numerics) against 50-digit arithmetic (mpmath): the checker of
the checker.  The reference's tests for this path hold one two-dimensional literal and otherwise compare with dense `exp`
computed in the same run in double precision, so this is the independent ground truth the image offers
(oracle/qp_oracle_mp.py; the full sweeps: tools/oracle_mpmath_pin.py -> profiles/r04/oracle_mpmath_pin.txt).  CPU only.

What is established here:
* `cheby_coeffs` (src/cheby.jl:25-39): the number of coefficients from the oracle (scipy `jv`: AMOS), from the library
  (`qp_cheby_coeffs`: glibc `jn`, the same Sun msun code as the openlibm `jn` that SpecialFunctions.besselj(::Int, ::Float64)
  calls in the reference) and from mpmath.besselj agree EXACTLY, for the configs' alpha and random alpha in (0, 300); the
  values agree to 5e-14 (oracle) / 5e-15 (library) absolutely -- the library is the closer one --, and the summed
  deviation, which bounds the effect on |psi>, stays below 1e-11;
* `cheby!` and `newton!` of the oracle against `exp(-i H dt) psi` in 50 digits at N <= 32 (Hermitian and a Liouvillian):
  < 1e-13, three orders inside the 1e-10 parity bar;
* the Leja ordering of fixture F4's first restart is the 50-digit ordering, pick for pick, with the margins by which each
  pick won; the divided differences of F4 differ from the 50-digit ones by O(1) RELATIVELY in the last (tiny)
  coefficients -- divided differences are ill-conditioned in the coefficients; the reference does the same arithmetic --
  while the interpolation polynomial they define agrees to 1e-14 on the spectrum: what src/newton.jl:176-214 relies on.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mp = pytest.importorskip("mpmath")
from oracle import qp_oracle as qo  # noqa: E402
from oracle import qp_oracle_mp as qm  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _to_np(v):
    return np.array([complex(x) for x in v])


def _alphas():
    rng = np.random.default_rng(20261004)
    # the configs: alpha = 10 (C2/C4/C5, dt = 1), 2 and 50, 5 (C1), and test_cheby.jl's N = 1000 case
    # (Delta ~ 1000 for Hermitian(rand) with dt = 0.5 => alpha ~ 250; 267 / 268 coefficients)
    return [10.0, 2.0, 50.0, 5.0, 250.0, 0.0, 1e-3, 299.999] + list(rng.uniform(0.0, 300.0, 72))


def test_cheby_coeffs_count_and_values_against_mpmath():
    worst = {"oracle": 0.0, "library": 0.0}
    worst_sum = {"oracle": 0.0, "library": 0.0}
    for alpha in _alphas():
        Delta, dt = 2.0 * alpha, 1.0
        exact = qm.cheby_coeffs(Delta, dt)
        got = {"oracle": qo.cheby_coeffs(Delta, dt), "library": L.cheby_coeffs(Delta, dt)}
        for who, c in got.items():
            assert len(c) == len(exact), f"alpha = {alpha}: {who} has {len(c)} coefficients, 50-digit arithmetic {len(exact)}"
            dev = [abs(float(mp.mpf(float(ck)) - ek)) for ck, ek in zip(c, exact)]
            worst[who] = max(worst[who], max(dev))
            worst_sum[who] = max(worst_sum[who], sum(dev))
    assert worst["oracle"] < 5e-14 and worst["library"] < 5e-15, worst
    # |psi - psi_exact| <= sum_k |a_k - a_k^exact| (the Chebyshev polynomials of the normalised H are bounded by 1)
    assert worst_sum["oracle"] < 1e-11 and worst_sum["library"] < 1e-12, worst_sum


def test_cheby_coeffs_truncation_rule_keeps_the_first_small_coefficient():
    """src/cheby.jl:32-37 appends, then tests: the last coefficient is the first one <= limit, the one before is above."""
    for alpha in (10.0, 50.0, 250.0):
        c = qm.cheby_coeffs(2.0 * alpha, 1.0)
        assert abs(c[-1]) <= mp.mpf(1e-12) < abs(c[-2])
    assert len(qm.cheby_coeffs(20.0, 1.0)) == 32 == len(qo.cheby_coeffs(20.0, 1.0))     # the headline's 31 mat-vecs


@pytest.mark.parametrize("N", [2, 5, 16, 32])
def test_oracle_cheby_and_newton_against_50_digit_expm_hermitian(N):
    rng = np.random.default_rng(100 + N)
    H = synth.dense_hermitian(N, rho=4.0, rng=rng)
    ev = np.linalg.eigvalsh(H)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    for dt in (0.7, -0.7):
        exact = _to_np(qm.expm_apply(H, psi0, dt))
        wrk = qo.ChebyWrk(psi0, ev[-1] - ev[0] + 0.2, ev[0] - 0.1, abs(dt))
        assert np.linalg.norm(qo.cheby(psi0.copy(), H, dt, wrk) - exact) < 1e-13
        if N > 3:
            nw = qo.NewtonWrk(psi0, m_max=min(10, N - 1))
            assert np.linalg.norm(qo.newton(psi0.copy(), H, dt, nw) - exact) < 1e-13


@pytest.mark.parametrize("n", [4, 5])
def test_oracle_newton_against_50_digit_expm_liouvillian(n):
    """The non-Hermitian case of test/test_newton.jl:116-125: a Liouvillian, default func (exp(-i L dt)) and func = exp."""
    Lm = synth.liouvillian_tridiag(n).toarray()
    rho0 = synth.random_state(n * n)
    exact = _to_np(qm.expm_apply(Lm, rho0, 0.5))
    got = qo.newton(rho0.copy(), Lm, 0.5, qo.NewtonWrk(rho0, m_max=10))
    assert np.linalg.norm(got - exact) < 1e-13
    exact_e = _to_np(qm.expm_apply(-1j * Lm, rho0, 0.5, func="exp"))
    got_e = qo.newton(rho0.copy(), -1j * Lm, 0.5, qo.NewtonWrk(rho0, m_max=10), func=np.exp)
    assert np.linalg.norm(got_e - exact_e) < 1e-13
    assert np.linalg.norm(exact_e - exact) < 1e-30 * 1e15      # the two forms are the same map


def test_f4_leja_order_is_the_50_digit_order():
    """Fixture F4 (first restart of the N = 256 Liouvillian Newton step): 20 Leja points out of the 210 accumulated Ritz
    values (src/newton.jl:297-300, :97-148).  The library's selection is checked against the oracle's elsewhere
    (tests/test_cabi_host.py); here the oracle's against exact products."""
    d = np.load(os.path.join(GOLD, "F4_newton_liouvillian_n256.npz"))
    chosen, margins = qm.extend_leja([], 0, list(d["first_ritz"]), 20)
    assert [complex(c) for c in chosen] == [complex(z) for z in d["leja"][:20]]
    # no pick was a near-tie: the runner-up's product is at least 1e-6 (relative) below -- far above double rounding
    assert min(m for m in margins if m is not None) > 1e-6
    # and the library's host routine makes the same picks from the same candidates
    leja = np.zeros(64, dtype=np.complex128)
    out, n = L.extend_leja(leja, 0, d["first_ritz"].copy(), 20)
    assert n == 20 and np.array_equal(out[:20], d["leja"][:20])


def test_f4_divided_differences_against_50_digits():
    d = np.load(os.path.join(GOLD, "F4_newton_liouvillian_n256.npz"))
    leja, a, radius = d["leja"][:20], d["a"][:20], float(d["radius"])
    exact = qm.extend_newton_coeffs(leja, radius)
    # the leading coefficients (those that carry the result) agree to working precision ...
    scale = max(abs(x) for x in exact)
    dev = [abs(mp.mpc(complex(x).real, complex(x).imag) - y) for x, y in zip(a, exact)]
    assert max(float(e / abs(y)) for e, y in zip(dev[:6], exact[:6])) < 1e-14
    # ... the error then grows by about a digit per order (divided differences are ill-conditioned in the coefficients:
    # here 7e-10 of the largest coefficient at k = 19, which is O(1) of that coefficient's own 2e-10 -- the reference does
    # the same arithmetic, and its convergence test reads this last coefficient, src/newton.jl:330-334: the library has to
    # reproduce the oracle's operation order, not the exact value, to take the same restart decisions) ...
    assert max(float(e / scale) for e in dev) < 1e-8
    assert abs(exact[-1]) < 1e-9 * scale and float(dev[-1] / abs(exact[-1])) < 1e2
    # ... and the polynomial both sets define is the same on the spectrum's hull, and interpolates f there
    pts = list(d["first_ritz"][::7]) + list(leja)
    for x in pts:
        p_double = qm.newton_polynomial(a, leja, radius, x)
        p_exact = qm.newton_polynomial(exact, leja, radius, x)
        f = mp.exp(mp.mpc(0, -1) * mp.mpc(complex(x).real, complex(x).imag))
        assert abs(p_double - p_exact) < 1e-14
        assert abs(p_exact - f) < 1e-13
    # the library's host routine (C++ complex arithmetic, libm's exp) runs the same recurrence: its coefficients sit in the
    # same error envelope around the exact ones -- they differ from the ORACLE's by as much as either differs from the truth
    # (5e-10 of the scale at k = 19: one ulp in exp(-i z_k) is amplified like any other rounding) -- and define the same
    # polynomial.  This is why parity of `newton!` is stated on |psi> (1e-10) and on the restart count, not on `a`.
    a_lib = np.zeros(64, dtype=np.complex128)
    a_lib, n_a = L.extend_newton_coeffs(a_lib, 0, leja.copy(), None, 20, radius)
    assert n_a == 20
    dev_lib = [abs(mp.mpc(complex(x).real, complex(x).imag) - y) for x, y in zip(a_lib[:20], exact)]
    assert max(float(e / abs(y)) for e, y in zip(dev_lib[:6], exact[:6])) < 1e-14
    assert max(float(e / scale) for e in dev_lib) < 1e-8
    for x in pts:
        assert abs(qm.newton_polynomial(a_lib[:20], leja, radius, x) - qm.newton_polynomial(exact, leja, radius, x)) < 1e-14
