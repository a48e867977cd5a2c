#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/*.npz.

The reference (Julia) cannot be run in the build container and holds no stored vectors for
this path, so these fixtures are produced by the NumPy oracle (oracle/qp_oracle.py), which is
itself pinned by the reference's known-answer tests (tests/test_oracle_kat.py).  They freeze
inputs AND expected outputs, so that (a) a later change of the oracle cannot silently move
the parity target and (b) the GPU path is checked against committed numbers, not only against
whatever the oracle computes at test time.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def optomech():
    """test/optomech.jl:1-44 (deterministic 55-dimensional sparse H)."""
    w_mech, g, eta = 10.0, 1.0, 2.0
    Delta = -w_mech
    N_cav, N_mech = 4, 10
    destroy = lambda N: sp.diags([np.sqrt(np.arange(1, N + 1)).astype(complex)], [1], format="csr")  # noqa: E731
    ident = lambda N: sp.identity(N + 1, dtype=complex, format="csr")  # noqa: E731
    a = sp.kron(destroy(N_cav), ident(N_mech)).tocsr()
    at = a.conj().T.tocsr()
    b = sp.kron(ident(N_cav), destroy(N_mech)).tocsr()
    bt = b.conj().T.tocsr()
    H = ((-Delta * at @ a + eta * (a + at)) + w_mech * bt @ b + (-g * (bt + b) @ at @ a)).tocsr()
    H.sort_indices()
    psi0 = np.zeros((N_cav + 1) * (N_mech + 1), dtype=complex)
    psi0[2] = 1.0
    return H, psi0


def csr_fields(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return dict(rowptr=A.indptr.astype(np.int64), col=A.indices.astype(np.int32), vals=A.data.astype(np.complex128),
                n=np.int64(A.shape[0]))


def main():
    # F1: optomech, 10 steps of dt = 0.2, Cheby (manual range from exact eigenvalues) and Newton (m_max = 10)
    H, psi0 = optomech()
    ev = np.linalg.eigvalsh(H.toarray())
    tlist = np.arange(0, 2.0 + 1e-9, 0.2)
    E_min, E_max = float(np.floor(ev[0]) - 1), float(np.ceil(ev[-1]) + 1)
    out_c, st_c = qo.propagate(psi0, H, tlist, "cheby", storage=True, E_min=E_min, E_max=E_max)
    out_n, st_n = qo.propagate(psi0, H, tlist, "newton", storage=True, m_max=10)
    np.savez_compressed(os.path.join(HERE, "F1_optomech.npz"), **csr_fields(H), psi0=psi0, tlist=tlist, E_min=E_min,
                        E_max=E_max, cheby_states=st_c, newton_states=st_n)

    # F3: the C2 generator at N = 256 (offsets scaled), 3 forward steps + 1 backward, dt = 1, range [-10, 10]
    N = 256
    offs = (1, 2, 3, 4, 16, 32, 48, 64)
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offs)
    Hs = synth.to_scipy(rp, col, vals, N)
    psi0 = synth.random_state(N)
    wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
    psi = psi0.copy()
    states = [psi0.copy()]
    for dt in (1.0, 1.0, 1.0, -1.0):
        qo.cheby(psi, Hs, dt, wrk)
        states.append(psi.copy())
    np.savez_compressed(os.path.join(HERE, "F3_cheby_c2_n256.npz"), rowptr=rp, col=col, vals=vals, n=np.int64(N),
                        offsets=np.array(offs), psi0=psi0, coeffs=wrk.coeffs, Delta=20.0, E_min=-10.0,
                        dts=np.array([1.0, 1.0, 1.0, -1.0]), states=np.stack(states, axis=1))

    # F4: Liouvillian n = 16 (N = 256), Newton m_max = 20, dt = 0.5: result + intermediates of the first restart
    Lm = synth.liouvillian_tridiag(16)
    rho0 = synth.random_state(Lm.shape[0])
    nw = qo.NewtonWrk(rho0, m_max=20)
    rho = qo.newton(rho0.copy(), Lm, 0.5, nw, record=True)
    t0 = nw.trace[0]
    np.savez_compressed(os.path.join(HERE, "F4_newton_liouvillian_n256.npz"), **csr_fields(Lm), rho0=rho0, dt=0.5,
                        m_max=np.int64(20), result=rho, restarts=np.int64(nw.restarts), n_a=np.int64(nw.n_a),
                        radius=nw.radius, a=nw.a[:nw.n_a], leja=nw.leja[:nw.n_a], first_Hess=t0["Hess"],
                        first_ritz=t0["ritz"], first_P=t0["P"], first_R=t0["R"], first_beta=t0["beta"])

    # F5: Ritz values / specrange with an explicit start vector (Hermitian sparse, N = 300)
    rng = np.random.default_rng(2026)
    Hh = synth.sparse_random(300, 0.05, rho=10.0, hermitian=True, rng=rng)
    st = rng.random(300) * np.exp(2j * np.pi * rng.random(300))
    st /= np.linalg.norm(st)
    ritz = qo.ritzvals(Hh, st, 20, 60, prec=1e-3)
    lo, hi = qo.specrange(Hh, "arnoldi", state=st, prec=1e-4)
    np.savez_compressed(os.path.join(HERE, "F5_specrange_n300.npz"), **csr_fields(Hh), state=st, ritz=ritz,
                        E_min=lo, E_max=hi)

    # F6: index work -- Julia-style CSC (1-based Int64) -> CSR, and row partitions
    A = synth.sparse_random(97, 0.08, rng=np.random.default_rng(7)).tocsc()
    A.sort_indices()
    rowptr, col, vals = qo.csc_to_csr(97, 97, A.indptr + 1, A.indices + 1, A.data)
    lens = np.random.default_rng(8).integers(0, 9, 1000)
    rp1000 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "F6_index_work.npz"), colptr=(A.indptr + 1).astype(np.int64),
                        rowval=(A.indices + 1).astype(np.int64), nzval=A.data, rowptr=rowptr, col=col, vals=vals,
                        rp1000=rp1000, parts_rows_8=qo.partition_rows(rp1000, 8), parts_nnz_8=qo.partition_rows(rp1000, 8, "nnz"),
                        parts_rows_3=qo.partition_rows(rp1000, 3), parts_nnz_3=qo.partition_rows(rp1000, 3, "nnz"))
    # F2: config C1 -- N = 128 dense complex Hermitian, Cheby, 200 steps (alpha ~ 5), checkpoints every 50 steps;
    #     the exact exponential is the independent check (test/test_cheby.jl:24-47 pattern)
    rng = np.random.default_rng(128)
    N = 128
    Hd = synth.dense_hermitian(N, rho=10.0, rng=rng)
    evd = np.linalg.eigvalsh(Hd)
    E_min, E_max = float(np.floor(evd[0]) - 1), float(np.ceil(evd[-1]) + 1)
    dt = 10.0 / (E_max - E_min)                     # alpha = Delta dt / 2 = 5
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    wrk = qo.ChebyWrk(psi0, E_max - E_min, E_min, dt)
    psi = psi0.copy()
    cps = []
    for k in range(200):
        qo.cheby(psi, Hd, dt, wrk)
        if (k + 1) % 50 == 0:
            cps.append(psi.copy())
    np.savez_compressed(os.path.join(HERE, "F2_cheby_c1_dense128.npz"), H=Hd, psi0=psi0, dt=dt, E_min=E_min, E_max=E_max,
                        n_coeffs=np.int64(wrk.n_coeffs), checkpoints=np.stack(cps, axis=1))

    # F7: Liouvillian superoperator applied to rho (src/generators.jl:473-631), n = 6, two Lindblad operators,
    #     both conventions, with a control coefficient: frozen targets for the sparse and the matrix-free operator
    rng = np.random.default_rng(7)
    n = 6
    H0 = synth.dense_hermitian(n, rho=2.0, rng=rng)
    H1 = synth.dense_hermitian(n, rho=1.0, rng=rng)
    cops = np.stack([0.4 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) for _ in range(2)])
    rho = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    vec = np.ascontiguousarray(rho.T).reshape(-1)    # column-major vec(rho)
    eps = 0.35
    out = {}
    for conv in ("TDSE", "LvN"):
        Ls = synth.ham_to_superop(sp.csr_matrix(H0), conv) + eps * synth.ham_to_superop(sp.csr_matrix(H1), conv)
        for A in cops:
            Ls = Ls + synth.lindblad_to_superop(sp.csr_matrix(A), conv)
        out[f"L_{conv}"] = Ls.toarray()
        out[f"Lrho_{conv}"] = Ls @ vec
    np.savez_compressed(os.path.join(HERE, "F7_liouvillian_n6.npz"), H0=H0, H1=H1, c_ops=cops, rho=rho, eps=eps, **out)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
