#!/usr/bin/env python3
"""Fuzz of newton! on the MULTI-LAUNCH Arnoldi path (more stored entries than the persistent kernel
takes: low-synchronisation Gram-Schmidt with the solve in the reduction's last workgroup, pipelined
Hessenberg eigenvalues, pre-folded Leja products, fused basis combination; the one-pass sweep of csrc/kernels_onepass.hip in a third of the cases) against the NumPy oracle:
random sparse systems, Hermitian or not, random Krylov size, time step sign, several steps.

    python tools/fuzz_newton_large.py [n_cases] [seed]"""
import os, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo   # checker
import qprop_amd.lib as L, qprop_amd.synth as synth

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = L.Context(0)
bad = 0
for case in range(ncases):
    rng = np.random.default_rng([seed, case])
    N = int(rng.integers(1200, 6000))
    dens = float(rng.uniform(14, 30)) / N
    herm = bool(rng.random() < 0.5)
    A = synth.sparse_random(N, dens, rho=float(rng.uniform(2, 8)), hermitian=herm, rng=rng)
    if not herm:      # keep the spectrum in the lower half plane (a dissipative generator), as for a Liouvillian
        A = (A - 1j * sp.identity(N) * float(abs(A).sum(axis=1).max()) * 0.5).tocsr()
    m_max = int(rng.choice([5, 8, 12, 20, 33, 50]))
    dt = float(rng.uniform(0.05, 0.6)) * (1 if (rng.random() < 0.7 or not herm) else -1)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    mode = int(rng.integers(0, 2))
    pipe = int(rng.integers(0, 2))
    onepass = int(rng.integers(0, 3))      # 0 never, 1 auto, 2 the one-pass sweep wherever an instance exists (m_max <= 20, mode 1)
    L.tuning_set("arnoldi_onepass", onepass)
    L.tuning_set("arnoldi_mode", mode)
    L.tuning_set("newton_pipeline", pipe)
    Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, sp.csr_matrix(A))])
    wrk = L.NewtonWrk(ctx, N, m_max=m_max)
    psi = L.State(ctx, data=psi0)
    ref = psi0.copy()
    owrk = qo.NewtonWrk(ref, m_max=m_max)
    nsteps = int(rng.integers(1, 4))
    try:
        for _ in range(nsteps):
            L.newton(psi, Op, dt, wrk, max_restarts=200)
            qo.newton(ref, A, dt, owrk, max_restarts=200)
        err = float(np.linalg.norm(psi.numpy() - ref))      # the parity bar of the test-suite: absolute, 1e-10
        ok = err < 1e-10
        msg = f"err {err:.2e} |psi|={np.linalg.norm(ref):.2e} restarts {wrk.restarts}/{owrk.restarts}"
    except Exception as e:   # noqa: BLE001
        ok, msg = False, f"{type(e).__name__}: {e}"
    if not ok:
        bad += 1
        print(f"case {case}: N={N} nnz={A.nnz} herm={herm} m_max={m_max} dt={dt:.3f} steps={nsteps} mode={mode} pipe={pipe} onepass={onepass}: {msg}",
              flush=True)
L.tuning_set("arnoldi_mode", 1)
L.tuning_set("arnoldi_onepass", 1)
L.tuning_set("newton_pipeline", 1)
print(f"{ncases} cases, {bad} bad")
sys.exit(1 if bad else 0)
