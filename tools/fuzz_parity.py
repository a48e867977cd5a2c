#!/usr/bin/env python3
"""Fuzz the device path against the NumPy oracle (test infrastructure: oracle/ is the checker):
random small systems -- size, sparsity, real / complex, Hermitian (Cheby + Newton) or not
(Newton), controls, time grid, forward / backward, storage, observables -- through
propagate() with every default (persistent kernels, formats, real copy, stencil ...).
Prints every case whose deviation exceeds 1e-10 and exits non-zero if there is one.

    python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.propagator as P  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None      # re-run one case verbosely
    ctx = L.Context(0)
    bad = 0
    for case in range(ncases):
        if only is not None and case != only:
            continue
        rng = np.random.default_rng([seed, case])
        n = int(rng.choice([2, 3, 5, 8, 17, 33, 64, 65, 100, 130, 200, 300]))
        dense = bool(rng.random() < 0.4) or n < 8
        real = bool(rng.random() < 0.4)
        method = "cheby" if rng.random() < 0.5 else "newton"
        if method == "newton" and n < 4:
            n = 5
        hermitian = method == "cheby" or rng.random() < 0.5
        if not hermitian:
            dense = True      # sparse random non-Hermitian draws are often defective (eigenvector condition 1e30):
                              # there the reference algorithm itself is unstable to 1e-15 perturbations
        ncontrols = int(rng.integers(0, 3))
        backward = bool(rng.random() < 0.3)
        nt = int(rng.integers(3, 12))

        def mat(rho):
            if dense:
                A = synth.dense_hermitian(n, rho=rho, rng=rng) if hermitian else synth.dense_nonhermitian(n, rho=rho, rng=rng)
                return A.real.astype(complex) if real else A
            A = synth.sparse_random(n, min(1.0, float(rng.uniform(1.5, 8.0)) / n), rho=rho, hermitian=hermitian, rng=rng)
            return sp.csr_matrix(A.real.astype(complex)) if real else A
        mats = [mat(2.0)] + [mat(0.5) for _ in range(ncontrols)]
        ctrls = [(lambda t, k=k: 0.6 * np.sin((k + 2) * t + 0.3)) for k in range(ncontrols)]
        if method == "newton" and rng.random() < 0.5:
            tlist = np.cumsum(np.concatenate([[0.0], 0.02 + 0.05 * rng.random(nt - 1)]))
        else:
            tlist = np.linspace(0, float(rng.uniform(0.2, 1.5)), nt)
        gen = P.hamiltonian(mats[0], *[(m, c) for m, c in zip(mats[1:], ctrls)]) if ncontrols else (mats[0],)
        ogen = qo.Generator(mats, ctrls) if ncontrols else mats[0]
        psi0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        psi0 /= np.linalg.norm(psi0)
        # Newton with a Krylov space of 3-4 on a non-normal H either converges within ~5 restarts or stalls
        # at the rounding floor (beta grows, |a_k| cannot fall below eps): keep the fuzz in the well-posed regime
        kw = dict(E_min=-5.0, E_max=5.0) if method == "cheby" else dict(m_max=int(rng.integers(3 if hermitian else 6, 10)))
        desc = f"case {case}: n={n} {method} dense={dense} real={real} herm={hermitian} controls={ncontrols} backward={backward} nt={nt}"
        # the oracle first: a case in which the algorithm itself gives up (e.g. max_restarts for a
        # strongly non-normal H and a small Krylov space) must give up on the device too
        try:
            ref, rst = qo.propagate(psi0, ogen, tlist, method, storage=True, backward=backward, **kw)
            oracle_exc = None
        except AssertionError as exc:
            oracle_exc = exc
        try:
            out, st = P.propagate(psi0, gen, tlist, method=method, storage=True, backward=backward, ctx=ctx, **kw)
            device_exc = None
        except L.QPError as exc:
            device_exc = exc
        if oracle_exc is not None or device_exc is not None:
            if (oracle_exc is None) != (device_exc is None):
                print("ONE-SIDED FAILURE", desc, "oracle:", oracle_exc, "device:", device_exc, flush=True)
                bad += 1
            continue
        scale = max(1.0, float(np.max(np.linalg.norm(rst, axis=0))))      # non-unitary dynamics can grow
        err = max(float(np.linalg.norm(out - ref)), float(np.max(np.linalg.norm(st - rst, axis=0)))) / scale
        O = mats[0]
        _, ev = P.propagate(psi0, gen, tlist, method=method, storage=True, backward=backward, observables=[O], ctx=ctx, **kw)
        O_ = O.toarray() if sp.issparse(O) else O
        want = np.array([np.vdot(rst[:, i], O_ @ rst[:, i]) for i in range(len(tlist))])
        err = max(err, float(np.max(np.abs(ev[0] - want))) / scale ** 2)
        if only is not None:
            print(desc, "err", err, "scale", scale, "tlist", tlist)
        if not err < 1e-10:
            print(f"MISMATCH {err:.3e}", desc, flush=True)
            bad += 1
    print(f"{ncases} cases, {bad} bad")
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
