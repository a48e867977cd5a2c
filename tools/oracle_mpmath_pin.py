#!/usr/bin/env python3
"""The full sweeps behind tests/test_oracle_mpmath.py (which runs a 80-alpha sample in the CPU suite): oracle (scipy `jv`),
library host numerics (glibc `jn`) and 50-digit arithmetic (mpmath) side by side.  CPU only, ~3 min.

    python tools/oracle_mpmath_pin.py > profiles/r04/oracle_mpmath_pin.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpmath as mp  # noqa: E402
from oracle import qp_oracle as qo  # noqa: E402
from oracle import qp_oracle_mp as qm  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def main():
    t0 = time.time()
    rng = np.random.default_rng(20261004)
    alphas = [10.0, 2.0, 50.0, 5.0, 250.0] + list(rng.uniform(0.0, 300.0, 1000))
    print(f"# cheby_coeffs (src/cheby.jl:25-39), limit 1e-12: {len(alphas)} values of alpha = Delta dt / 2 (the configs' 10, 2, 50, 5, "
          f"test_cheby.jl's ~250, and 1000 uniform in (0, 300)); mpmath at {mp.mp.dps} digits")
    mism = {"oracle": 0, "library": 0}
    worst = {"oracle": (0.0, None), "library": (0.0, None)}
    worst_sum = {"oracle": (0.0, None), "library": (0.0, None)}
    nearest = (1.0, None)
    for alpha in alphas:
        exact = qm.cheby_coeffs(2.0 * alpha, 1.0)
        # how close the deciding coefficients are to the limit (a count can only differ if one is within rounding of it)
        for k in (-1, -2):
            gap = abs(float(abs(exact[k]) / mp.mpf(1e-12) - 1))
            if gap < nearest[0]:
                nearest = (gap, (alpha, len(exact) + k))
        for who, c in (("oracle", qo.cheby_coeffs(2.0 * alpha, 1.0)), ("library", L.cheby_coeffs(2.0 * alpha, 1.0))):
            if len(c) != len(exact):
                mism[who] += 1
                print(f"  COUNT differs: alpha = {alpha!r}: {who} {len(c)}, exact {len(exact)}")
            dev = [abs(float(mp.mpf(float(ck)) - ek)) for ck, ek in zip(c, exact)]
            if max(dev) > worst[who][0]:
                worst[who] = (max(dev), (alpha, int(np.argmax(dev))))
            if sum(dev) > worst_sum[who][0]:
                worst_sum[who] = (sum(dev), alpha)
    for who, src in (("oracle", "scipy.special.jv (AMOS)"), ("library", "glibc jn (Sun msun, as openlibm's jn behind SpecialFunctions.besselj)")):
        print(f"{who:8s} [{src}]: coefficient counts differing from exact: {mism[who]} of {len(alphas)}; "
              f"largest |a_k - exact| {worst[who][0]:.2e} (alpha {worst[who][1][0]:.3f}, k {worst[who][1][1]}); "
              f"largest sum_k |a_k - exact| {worst_sum[who][0]:.2e} (alpha {worst_sum[who][1]:.3f}) = bound on the effect on |psi>")
    print(f"closest approach of a deciding coefficient to the limit: | |a_k| / 1e-12 - 1 | = {nearest[0]:.2e} (alpha {nearest[1][0]:.3f}, k {nearest[1][1]})"
          f" -- double rounding is 1e-16: no count was decided by rounding")

    print("\n# exp(-i H dt) psi at N <= 32: oracle cheby! / newton! (double) against mpmath.expm (50 digits); |delta psi|_2")
    for N in (2, 3, 5, 8, 16, 24, 32):
        r = np.random.default_rng(100 + N)
        H = synth.dense_hermitian(N, rho=4.0, rng=r)
        ev = np.linalg.eigvalsh(H)
        psi0 = r.standard_normal(N) + 1j * r.standard_normal(N)
        psi0 /= np.linalg.norm(psi0)
        ex = np.array([complex(x) for x in qm.expm_apply(H, psi0, 0.7)])
        c = qo.cheby(psi0.copy(), H, 0.7, qo.ChebyWrk(psi0, ev[-1] - ev[0] + 0.2, ev[0] - 0.1, 0.7))
        line = f"Hermitian N = {N:2d}: cheby {np.linalg.norm(c - ex):.2e}"
        if N > 3:
            nw = qo.NewtonWrk(psi0, m_max=min(10, N - 1))
            line += f"   newton {np.linalg.norm(qo.newton(psi0.copy(), H, 0.7, nw) - ex):.2e} ({nw.restarts} restarts)"
        print(line)
    for n in (3, 4, 5):
        Lm = synth.liouvillian_tridiag(n).toarray()
        rho0 = synth.random_state(n * n)
        ex = np.array([complex(x) for x in qm.expm_apply(Lm, rho0, 0.5)])
        nw = qo.NewtonWrk(rho0, m_max=min(10, n * n - 1))
        print(f"Liouvillian N = {n * n:2d} (non-Hermitian): newton {np.linalg.norm(qo.newton(rho0.copy(), Lm, 0.5, nw) - ex):.2e}")

    print("\n# fixture F4, first restart: Leja ordering and divided differences in 50 digits (src/newton.jl:97-148, :176-214)")
    d = np.load(os.path.join(ROOT, "tests", "golden", "F4_newton_liouvillian_n256.npz"))
    leja, a, radius = d["leja"][:20], d["a"][:20], float(d["radius"])
    chosen, margins = qm.extend_leja([], 0, list(d["first_ritz"]), 20)
    same = sum(complex(c) == complex(z) for c, z in zip(chosen, leja))
    print(f"Leja picks identical to the oracle's: {same} of 20; smallest winning margin 1 - p_second / p_max = {min(m for m in margins if m is not None):.2e}")
    exact = qm.extend_newton_coeffs(leja, radius)
    a_lib = np.zeros(64, dtype=np.complex128)
    a_lib, _ = L.extend_newton_coeffs(a_lib, 0, leja.copy(), None, 20, radius)
    scale = max(abs(x) for x in exact)
    print(" k   |a_k| exact   |oracle - exact|/max|a|   |library - exact|/max|a|   |oracle - library|/max|a|")
    for k in range(20):
        eo = abs(mp.mpc(a[k].real, a[k].imag) - exact[k]) / scale
        el = abs(mp.mpc(a_lib[k].real, a_lib[k].imag) - exact[k]) / scale
        print(f"{k:2d}   {float(abs(exact[k])):.3e}     {float(eo):.2e}                 {float(el):.2e}                  {abs(a[k] - a_lib[k]) / float(scale):.2e}")
    pts = list(d["first_ritz"][::7]) + list(leja)
    po = max(float(abs(qm.newton_polynomial(a, leja, radius, x) - qm.newton_polynomial(exact, leja, radius, x))) for x in pts)
    pl = max(float(abs(qm.newton_polynomial(a_lib[:20], leja, radius, x) - qm.newton_polynomial(exact, leja, radius, x))) for x in pts)
    pf = max(float(abs(qm.newton_polynomial(exact, leja, radius, x) - mp.exp(mp.mpc(0, -1) * mp.mpc(complex(x).real, complex(x).imag)))) for x in pts)
    print(f"interpolation polynomial on {len(pts)} points of the Ritz hull: |p_oracle - p_exact| <= {po:.2e}, |p_library - p_exact| <= {pl:.2e}, "
          f"|p_exact - exp(-i z)| <= {pf:.2e}")
    print("reading: the coefficients' rounding error grows about a digit per order (divided differences); any two double implementations "
          "differ in the last coefficients by as much as either differs from the truth, the polynomial does not.  newton! parity is "
          "therefore stated on |psi> and on the restart count.")
    print(f"# {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
