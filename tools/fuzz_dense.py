#!/usr/bin/env python3
"""Fuzz the dense operator format (QP_FMT_DENSE, csrc/kernels_dense.hip) against NumPy / the oracle (test infrastructure):
random shapes (rectangular for mul!, ragged tile edges, inner dimensions that are no multiple of 4), real / complex values,
lazy sums with coefficients and scale, one state through cheby! and newton!, panels of 1 ... 100 states through the
matrix-core kernel (both tile widths), forward and backward.

    python tools/fuzz_dense.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def dense_matrix(ctx, A):
    nr, nc = A.shape
    return L.Matrix(ctx, nr, nc, np.arange(nr + 1, dtype=np.int64) * nc, np.tile(np.arange(nc, dtype=np.int32), nr),
                    np.ascontiguousarray(A, dtype=np.complex128).reshape(-1))


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = L.Context(0)
    bad = ill = 0
    kinds = {"mul": 0, "cheby": 0, "newton": 0, "panel": 0}
    for case in range(ncases):
        rng = np.random.default_rng([seed, case])
        kind = ["mul", "cheby", "newton", "panel"][int(rng.integers(0, 4))]
        real = bool(rng.random() < 0.3)
        kinds[kind] += 1
        err = 0.0
        if kind == "mul":
            nr, nc = int(rng.integers(1, 700)), int(rng.integers(1, 900))
            nterms = int(rng.integers(1, 4))
            mats = [rng.standard_normal((nr, nc)) + (0 if real else 1j) * rng.standard_normal((nr, nc)) for _ in range(nterms)]
            op = L.Operator(ctx, [dense_matrix(ctx, A) for A in mats], nterms - 1)
            assert op.format == L.FMT_DENSE
            coeffs = [complex(rng.standard_normal(), 0 if real else rng.standard_normal()) for _ in range(nterms - 1)]
            if coeffs:
                op.set_coeffs(coeffs)
            scale = complex(rng.standard_normal(), rng.standard_normal()) if rng.random() < 0.5 else 1.0
            op.set_scale(scale)
            Aeff = scale * (mats[0] + sum(c * A for c, A in zip(coeffs, mats[1:])))
            x = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
            y0 = rng.standard_normal(nr) + 1j * rng.standard_normal(nr)
            al, be = complex(rng.standard_normal(), rng.standard_normal()), complex(rng.standard_normal(), rng.standard_normal())
            Y = L.State(ctx, data=y0)
            op.mul(L.State(ctx, data=x), Y, al, be)
            ref = al * (Aeff @ x) + be * y0
            err = float(np.linalg.norm(Y.numpy() - ref) / max(1.0, np.linalg.norm(ref)))
            desc = f"mul {nr}x{nc} terms={nterms} real={real}"
        else:
            n = int(rng.integers(4, 500))
            H = synth.dense_hermitian(n, rho=4.0, rng=rng)
            if real:
                H = H.real.astype(np.complex128)
            if kind == "newton" and rng.random() < 0.5:
                H = synth.dense_nonhermitian(n, rho=3.0, rng=rng)
            op = L.Operator(ctx, [dense_matrix(ctx, H)])
            assert op.format == L.FMT_DENSE
            dt = float(rng.uniform(0.05, 0.6)) * (1 if rng.random() < 0.7 else -1)
            psi0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            psi0 /= np.linalg.norm(psi0)
            if kind == "newton":
                m = int(rng.integers(5, 12))
                if m >= n:
                    m = n - 1
                if m <= 2:
                    continue
                wrk = L.NewtonWrk(ctx, n, m_max=m)
                psi = L.State(ctx, data=psi0)
                owrk = qo.NewtonWrk(psi0, m_max=m)
                try:
                    ref = qo.newton(psi0.copy(), H, dt, owrk)
                except AssertionError:
                    continue
                desc = f"newton n={n} m={m} dt={dt:.3f} real={real} hermitian={bool(np.allclose(H, H.conj().T))}"
                try:
                    L.newton(psi, op, dt, wrk)
                    err = float(np.linalg.norm(psi.numpy() - ref))
                except L.QPError as e:      # the oracle converged: a device-side failure is a mismatch, not a crash of the fuzzer
                    # newton! decides convergence on beta |a_last| (src/newton.jl:370), and the last divided difference is
                    # rounding noise at its own scale (tests/test_oracle_mpmath.py): an input on which the ORACLE converges with less
                    # than a factor 2 to spare and diverges when that test is missed is decided by rounding, not by the device --
                    # counted as an ill-conditioned draw, everything else as a mismatch (and saved for a closer look)
                    knife_edge = False
                    try:
                        qo.newton(psi0.copy(), H, dt, qo.NewtonWrk(psi0, m_max=m), relerr=0.5e-12)
                    except AssertionError:
                        knife_edge = True
                    if knife_edge:
                        err = 0.0
                        ill += 1
                        print(f"ill-conditioned draw, case {case}: {desc}: the device did not converge ({e}); the oracle converges in "
                              f"{owrk.restarts} restarts at relerr = 1e-12 and fails at 0.5e-12", flush=True)
                    else:
                        err = float("inf")
                        desc += f" [{e}]; the oracle took {owrk.restarts} restarts"
                        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_dense_case{case}.npz"), H=H, psi0=psi0, dt=dt, m=m, ref=ref)
            else:
                ev = np.linalg.eigvalsh(H)
                Delta, E_min = ev[-1] - ev[0] + 0.5, ev[0] - 0.25
                if kind == "cheby":
                    wrk = L.ChebyWrk(ctx, n, Delta, E_min, abs(dt))
                    psi = L.State(ctx, data=psi0)
                    L.cheby(psi, op, dt, wrk, check_normalization=bool(rng.random() < 0.3))
                    ref = qo.cheby(psi0.copy(), H, dt, qo.ChebyWrk(psi0, Delta, E_min, abs(dt)))
                    err = float(np.linalg.norm(psi.numpy() - ref))
                    desc = f"cheby n={n} dt={dt:.3f} real={real}"
                else:
                    b = int(rng.choice([1, 2, 7, 8, 15, 16, 17, 31, 32, 33, 64, 100]))
                    states = rng.standard_normal((n, b)) + 1j * rng.standard_normal((n, b))
                    states /= np.linalg.norm(states, axis=0)
                    panel = L.State(ctx, data=states.reshape(-1))
                    wrk = L.ChebyWrk(ctx, n * b, Delta, E_min, abs(dt))
                    L.cheby_batched(panel, op, dt, wrk, b)
                    got = panel.numpy().reshape(n, b)
                    for s in range(b):
                        ref = qo.cheby(states[:, s].copy(), H, dt, qo.ChebyWrk(states[:, s], Delta, E_min, abs(dt)))
                        err = max(err, float(np.linalg.norm(got[:, s] - ref)))
                    desc = f"panel n={n} b={b} dt={dt:.3f} real={real}"
        if not err < 1e-10:
            print(f"MISMATCH {err:.3e} case {case}: {desc}", flush=True)
            bad += 1
    print(f"{ncases} cases ({kinds}), {bad} bad" + (f", {ill} ill-conditioned draw(s) of newton! (see above)" if ill else ""))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
