"""Pin the NumPy oracle against the reference's own known-answer tests.

The reference holds no stored golden vectors for this path (its tests compare
against dense `exp`, `eigvals` and analytic results computed in the same run), so
each reference test is restated here with the same recipe and the same tolerance.
Reference files are cited per test.  CPU only.
"""
import os
import sys

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def _rand_state(N, rng):
    psi = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    return psi / np.linalg.norm(psi)


def test_cheby_random_state():
    """test/test_cheby.jl:6-49: N=1000 `Hermitian(rand(ComplexF64,N,N))`, dt=0.5,
    coefficient count in {267, 268}, cheby! vs exp(-i H dt) to 1e-10."""
    rng = np.random.default_rng(1)
    N = 1000
    X = rng.random((N, N)) + 1j * rng.random((N, N))
    H = np.triu(X) + np.triu(X, 1).conj().T   # Julia Hermitian(X): upper triangle
    H[np.diag_indices(N)] = H[np.diag_indices(N)].real
    dt = 0.5
    psi0 = rng.random(N) + 1j * rng.random(N)
    psi0 /= np.linalg.norm(psi0)
    evals, V = np.linalg.eigh(H)
    expected = V @ (np.exp(-1j * evals * dt) * (V.conj().T @ psi0))
    E_min = evals[0]
    Delta = evals[-1] - evals[0]
    a = qo.cheby_coeffs(Delta, dt)
    assert len(a) in (267, 268)
    wrk = qo.ChebyWrk(psi0, Delta, E_min, dt)
    psi = psi0.copy()
    qo.cheby(psi, H, dt, wrk)
    assert np.linalg.norm(psi - expected) < 1e-10
    assert wrk.n_matvec == len(a) - 1


def test_cheby_backward_is_inverse():
    """src/cheby.jl:158-162, :211 sign conventions: dt<0 undoes dt>0."""
    rng = np.random.default_rng(2)
    H = synth.dense_hermitian(64, rho=3.0, rng=rng)
    ev = np.linalg.eigvalsh(H)
    psi0 = _rand_state(64, rng)
    wrk = qo.ChebyWrk(psi0, ev[-1] - ev[0], ev[0], 0.3)
    psi = psi0.copy()
    qo.cheby(psi, H, 0.3, wrk)
    qo.cheby(psi, H, -0.3, wrk)
    assert np.linalg.norm(psi - psi0) < 1e-12


def test_cheby_check_normalization_trips():
    """src/cheby.jl:194-200: too small a spectral radius must assert."""
    rng = np.random.default_rng(3)
    H = synth.dense_hermitian(64, rho=10.0, rng=rng)
    psi0 = _rand_state(64, rng)
    wrk = qo.ChebyWrk(psi0, 2.0, -1.0, 0.1)   # true range ~[-10, 10]
    with pytest.raises(AssertionError, match="Incorrect normalization"):
        qo.cheby(psi0.copy(), H, 0.1, wrk, check_normalization=True)


def test_newton_random_hermitian():
    """test/test_newton.jl:7-67: N=1000 Hermitian rho=10, dt=0.5, m_max=5,
    max_restarts=200; norm preserved; vs dense exp to 1e-10."""
    rng = np.random.default_rng(4)
    N = 1000
    H = synth.dense_hermitian(N, rho=10.0, rng=rng)
    psi0 = _rand_state(N, rng)
    ev, V = np.linalg.eigh(H)
    expected = V @ (np.exp(-1j * ev * 0.5) * (V.conj().T @ psi0))
    wrk = qo.NewtonWrk(psi0, m_max=5)
    psi = psi0.copy()
    qo.newton(psi, H, 0.5, wrk, max_restarts=200)
    assert abs(np.linalg.norm(psi) - 1) < 1e-10
    assert np.linalg.norm(psi - expected) < 1e-10
    assert wrk.restarts > 0


def test_newton_random_nonhermitian():
    """test/test_newton.jl:70-127: N=1000 non-Hermitian rho=10, m_max=50."""
    rng = np.random.default_rng(5)
    N = 1000
    H = synth.dense_nonhermitian(N, rho=10.0, rng=rng)
    psi0 = _rand_state(N, rng)
    expected = sla.expm(-1j * H * 0.5) @ psi0
    wrk = qo.NewtonWrk(psi0, m_max=50)
    psi = psi0.copy()
    qo.newton(psi, H, 0.5, wrk, max_restarts=200)
    assert np.linalg.norm(psi - expected) < 1e-10


def test_newton_sparse_liouvillian_custom_func():
    """test/test_newton.jl:130-177: sparse L of dimension 32^2, density 0.5,
    m_max=50, func = exp, max_restarts=20, vs exp(Array(L*dt)) to 1e-10."""
    rng = np.random.default_rng(6)
    N = 32
    L = synth.sparse_random(N * N, 0.5, rho=10.0, rng=rng)
    psi0 = _rand_state(N, rng)
    rho0 = np.outer(psi0, psi0.conj()).reshape(-1, order="F")
    assert abs(np.trace(rho0.reshape(N, N, order="F")) - 1) < 1e-12
    expected = sla.expm(L.toarray() * 0.5) @ rho0
    wrk = qo.NewtonWrk(rho0, m_max=50)
    rho = rho0.copy()
    qo.newton(rho, L, 0.5, wrk, max_restarts=20, func=np.exp)
    assert np.linalg.norm(rho - expected) < 1e-10


def test_newton_eigenstate_shortcut():
    """src/newton.jl:289-295: an eigenstate returns after one Arnoldi step."""
    rng = np.random.default_rng(7)
    H = synth.dense_hermitian(50, rho=4.0, rng=rng)
    ev, V = np.linalg.eigh(H)
    psi = V[:, 3].copy()
    wrk = qo.NewtonWrk(psi, m_max=10)
    qo.newton(psi, H, 0.7, wrk)
    assert wrk.restarts == 0
    assert np.linalg.norm(psi - np.exp(-1j * ev[3] * 0.7) * V[:, 3]) < 1e-12


def test_ritzvals_nonhermitian():
    """test/test_specrad.jl:14-45: ritzvals(X, psi, 180, 200; prec=1e-5) vs the
    (Re,Im)-sorted eigvals extremes, relative 1 %."""
    rng = np.random.default_rng(8)
    N = 1000
    X = synth.dense_nonhermitian(N, rho=10.0, rng=rng)
    psi = _rand_state(N, rng)
    ritz = qo.ritzvals(X, psi, 180, 200, prec=1e-5)
    ev = qo._eigvals_sorted(X)
    assert abs(ev[0] - ritz[0]) / abs(ev[0]) < 0.01
    assert abs(ev[-1] - ritz[-1]) / abs(ev[-1]) < 0.01


def test_ritzvals_hermitian():
    """test/test_specrad.jl:47-77: ritzvals(H, psi, 20, 60; prec=1e-3), 2 %."""
    rng = np.random.default_rng(9)
    N = 1000
    H = synth.dense_hermitian(N, rho=10.0, rng=rng)
    psi = _rand_state(N, rng)
    ritz = qo.ritzvals(H, psi, 20, 60, prec=1e-3)
    ev = np.linalg.eigvalsh(H)
    assert abs(ev[0] - ritz[0]) / abs(ev[0]) < 0.02
    assert abs(ev[-1] - ritz[-1]) / abs(ev[-1]) < 0.02


def test_specrange_methods():
    """test/test_specrad.jl:80-144."""
    rng = np.random.default_rng(10)
    N = 1000
    H = synth.sparse_random(N, 0.1, rho=10.0, hermitian=True, rng=rng)
    ev = np.linalg.eigvalsh(H.toarray())
    Delta = ev[-1] - ev[0]
    psi = _rand_state(N, rng)
    E_min, E_max = qo.specrange(H, "arnoldi", state=psi, prec=1e-4)
    assert ev[0] - 0.05 * Delta <= E_min <= ev[0]
    assert ev[-1] <= E_max < ev[-1] + 0.05 * Delta
    E_min, E_max = qo.specrange(H, "diag")
    assert abs(ev[0] - E_min) < 1e-12 and abs(ev[-1] - E_max) < 1e-12
    with pytest.raises(KeyError):
        qo.specrange(H, "manual")
    with pytest.raises(KeyError):
        qo.specrange(H, "manual", E_min=-1.0)
    assert qo.specrange(H, "manual", E_min=-10, E_max=10) == (-10.0, 10.0)
    E_min, E_max = qo.specrange(H, "auto", state=psi)
    assert ev[0] - 0.05 * Delta <= E_min <= ev[0]
    assert ev[-1] <= E_max < ev[-1] + 0.05 * Delta
    assert qo.specrange(H, "auto", E_min=-10, E_max=10) == (-10.0, 10.0)


def test_cheby_init_prop_spectral_envelope():
    """test/test_specrad.jl:147-223 (manual range and specrange_buffer):
    E in [-10,10] -> E_min ~ -10.1, Delta ~ 20.2; buffer 0.1 -> -11, 22."""
    rng = np.random.default_rng(11)
    H = synth.dense_hermitian(40, rho=5.0, rng=rng)
    psi = _rand_state(40, rng)
    tlist = np.linspace(0, 1, 11)
    p = qo.init_prop(psi, H, tlist, "cheby", E_min=-10, E_max=10)
    assert abs(p.wrk.E_min - (-10.1)) < 1e-12 and abs(p.wrk.Delta - 20.2) < 1e-12
    p = qo.init_prop(psi, H, tlist, "cheby", E_min=-10, E_max=10, specrange_buffer=0.1)
    assert abs(p.wrk.E_min - (-11.0)) < 1e-12 and abs(p.wrk.Delta - 22.0) < 1e-12


def test_tls_rabi_analytic():
    """test/test_propagate.jl:74-150: two-level Rabi flip, Cheby forward and
    backward vs analytic result to 1e-12."""
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    sz = np.array([[1, 0], [0, -1]], dtype=complex)
    Omega = 1.0
    T = np.pi / (2 * Omega)              # quarter Rabi cycle
    tlist = np.linspace(0, T, 101)
    gen = qo.Generator([0.0 * sz, 0.5 * sx], [lambda t: Omega])
    psi0 = np.array([1, 0], dtype=complex)
    # exp(-i (Omega/2) sx T) |0> = cos(pi/4)|0> - i sin(pi/4)|1>
    expected = np.array([1 / np.sqrt(2), -1j / np.sqrt(2)])
    out = qo.propagate(psi0, gen, tlist, "cheby")
    assert np.linalg.norm(out - expected) < 1e-12
    back = qo.propagate(out, gen, tlist, "cheby", backward=True)
    assert np.linalg.norm(back - psi0) < 1e-12


def test_tls_rabi_verbatim():
    """test/test_propagate.jl:10-71 and :74-150 with the reference's literals: H = [[0, 0.5], [0.5, 0]] given
    as the tuple generator `(H,)`, tlist = range(0, 1.5 pi, length = 101) -- a 3 pi / 2 pulse --, expected
    [-1/sqrt 2, -i/sqrt 2] ("note the phases"), forward and backward to 1e-12, the stored populations and the
    forward / backward storage arrays (method Cheby, as :110-112 and :138-147)."""
    psi0 = np.array([1, 0], dtype=complex)
    H = np.array([[0, 0.5], [0.5, 0]], dtype=complex)
    tlist = np.linspace(0, 1.5 * np.pi, 101)
    generator = (H,)
    out, storage = qo.propagate(psi0, generator, tlist, "cheby", storage=True)
    expected = np.array([-1 / np.sqrt(2), -1j / np.sqrt(2)])
    assert np.linalg.norm(out - expected) < 1e-12                       # :105, :112
    pop0 = np.abs(storage[0, :]) ** 2
    assert abs(pop0[-1] - 0.5) < 1e-8 and abs(pop0[0] - 1.0) < 1e-15    # :44 (isapprox)
    back, storage_bw = qo.propagate(out, generator, tlist, "cheby", backward=True, storage=True)
    assert np.linalg.norm(back - psi0) < 1e-12                          # :66, :147
    assert abs(abs(storage_bw[0, 0]) ** 2 - 1.0) < 1e-8                 # :67
    assert np.linalg.norm(storage - storage_bw) < 1e-12                 # :69


def _optomech():
    """test/optomech.jl:1-44 restated (deterministic, no RNG)."""
    w_mech, g, eta = 10.0, 1.0, 2.0
    Delta = -w_mech
    N_cav, N_mech = 4, 10

    def destroy(N):
        return sp.diags([np.sqrt(np.arange(1, N + 1)).astype(complex)], [1], format="csr")

    def ident(N):
        return sp.identity(N + 1, dtype=complex, format="csr")
    a = sp.kron(destroy(N_cav), ident(N_mech)).tocsr()
    at = a.conj().T.tocsr()
    b = sp.kron(ident(N_cav), destroy(N_mech)).tocsr()
    bt = b.conj().T.tocsr()
    H = (-Delta * at @ a + eta * (a + at)) + w_mech * bt @ b + (-g * (bt + b) @ at @ a)
    psi0 = np.zeros((N_cav + 1) * (N_mech + 1), dtype=complex)
    psi0[0 * (N_mech + 1) + 2] = 1.0
    return H.tocsr(), psi0


def test_optomech_newton_vs_cheby():
    """test/test_propagate.jl:153-163: tlist = 0:0.2:50, Newton norm to 1e-12,
    Newton vs Cheby to 1e-10 (specrange :auto -> :arnoldi since N=55 > 32)."""
    H, psi0 = _optomech()
    tlist = np.arange(0, 50 + 1e-9, 0.2)
    psi1 = qo.propagate(psi0, H, tlist, "newton")
    assert (np.linalg.norm(psi1) - 1.0) < 1e-12
    rng = np.random.default_rng(12)
    st = rng.random(55) * np.exp(2j * np.pi * rng.random(55))
    st /= np.linalg.norm(st)
    psi2 = qo.propagate(psi0, H, tlist, "cheby", specrange_state=st)
    assert np.linalg.norm(psi1 - psi2) < 1e-10
    ref = sla.expm(-1j * H.toarray() * 50.0) @ psi0
    assert np.linalg.norm(psi2 - ref) < 1e-9


def test_operator_mul():
    """test/test_operator_linalg.jl:30-64: mul!(phi, Op, psi, alpha, beta) for
    (1,0),(1,1),(2,1),(2,2); ScaledOperator; dot."""
    rng = np.random.default_rng(13)
    N = 30
    H0 = synth.dense_hermitian(N, rng=rng)
    H1 = synth.dense_hermitian(N, rng=rng)
    H2 = synth.dense_hermitian(N, rng=rng)
    psi = _rand_state(N, rng)
    phi0 = _rand_state(N, rng)
    for ops, coeffs in (([H0, H1, H2], [0.3, -1.2]), ([H1, H2], [0.7, 0.1 + 0.2j])):
        Op = qo.Operator(ops, coeffs)
        A = Op.toarray()
        for alpha, beta in ((1, 0), (1, 1), (2, 1), (2, 2)):
            out = Op.mul(psi, alpha, beta, C=phi0.copy())
            assert np.linalg.norm(out - (beta * phi0 + alpha * (A @ psi))) < 1e-12
        S = qo.ScaledOperator(0.5j, Op)
        assert np.linalg.norm(S.mul(psi, 2, 1, C=phi0.copy()) - (phi0 + 2 * 0.5j * (A @ psi))) < 1e-12
        assert abs(Op.dot(phi0, psi) - np.vdot(phi0, A @ psi)) < 1e-12


def test_csc_to_csr_bit_exact():
    """Index work at the boundary: Julia SparseMatrixCSC (1-based Int64) -> CSR."""
    rng = np.random.default_rng(14)
    A = synth.sparse_random(97, 0.08, rng=rng).tocsc()
    A.sort_indices()
    rowptr, col, vals = qo.csc_to_csr(97, 97, A.indptr + 1, A.indices + 1, A.data)
    B = A.tocsr()
    B.sort_indices()
    assert np.array_equal(rowptr, B.indptr) and np.array_equal(col, B.indices)
    assert np.array_equal(vals, B.data)
    assert rowptr.dtype == np.int64 and col.dtype == np.int32


def test_partition_rows():
    rowptr = np.arange(0, 17 * 4, 4)
    assert list(qo.partition_rows(rowptr, 3)) == [0, 6, 11, 16]
    b = qo.partition_rows(rowptr, 4, balance="nnz")
    assert list(b) == [0, 4, 8, 12, 16]


def test_synthetic_hermitian():
    rowptr, col, vals = synth.hermitian_offsets_csr(256, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    H = synth.to_scipy(rowptr, col, vals, 256)
    assert abs(H - H.conj().T).max() == 0
    assert np.all(np.diff(rowptr) == 16)
    assert np.max(np.abs(np.linalg.eigvalsh(H.toarray()))) <= 10.0
    # row-range generation is consistent with the full matrix
    r2, c2, v2 = synth.hermitian_offsets_csr(256, offsets=(1, 2, 3, 4, 16, 32, 48, 64),
                                             row_begin=100, row_end=180)
    assert np.array_equal(c2, col[rowptr[100]:rowptr[180]])
    assert np.array_equal(v2, vals[rowptr[100]:rowptr[180]])
    psi = synth.random_state(256)
    assert abs(np.linalg.norm(psi) - 1) < 1e-14
    assert np.array_equal(synth.random_state(256, row_begin=10, row_end=20), psi[10:20])


# ---- the C port (oracle/cheby_ref.c) is the checker of the full-size GPU tests: pin it too ---------------------------
# test_full_size_properties (C2), the C4 / C5 tests and bench.py's l2_diff compare the device with
# ref_c.cheby_csc / ChebyCsrOmp, and nothing compared THOSE with the NumPy oracle, a fixture or mpmath.

def _c_port_cases():
    import scipy.sparse as sp_
    rng = np.random.default_rng(20261004)
    f = np.load(os.path.join(ROOT, "tests", "golden", "F3_cheby_c2_n256.npz"))
    H3 = sp_.csr_matrix((f["vals"], f["col"], f["rowptr"]), shape=(256, 256)) if "vals" in f.files else None
    cases = []
    if H3 is not None:
        cases.append(("F3 N=256 (the C2 generator)", H3, f["psi0"], float(f["Delta"]), float(f["E_min"])))
    N = 4096
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    cases.append(("N=4096 banded", synth.to_scipy(rp, col, vals, N), synth.random_state(N), 20.0, -10.0))
    # ragged: rows of very different lengths (incl. empty ones), a real diagonal, Hermitian by construction
    Nr = 777
    lens = rng.integers(0, 40, Nr)
    lens[rng.integers(0, Nr, 60)] = 0
    rows = np.repeat(np.arange(Nr), lens)
    cols = rng.integers(0, Nr, rows.size)
    A = sp_.coo_matrix((rng.standard_normal(rows.size) + 1j * rng.standard_normal(rows.size), (rows, cols)), shape=(Nr, Nr)).tocsr()
    Hr = (A + A.getH()).tocsr() + sp_.diags(rng.standard_normal(Nr)).tocsr()
    Hr.sum_duplicates()
    Hr.sort_indices()
    bound = float(abs(Hr).sum(axis=1).max())
    cases.append(("ragged random Hermitian N=777", Hr, _rand_state(Nr, rng), 2.0 * bound, -bound))
    return cases


def test_c_port_matches_the_numpy_oracle():
    """ref_c.cheby_csc (serial CSC, the reference's operation order: src/cheby.jl:171-211) and ref_c.ChebyCsrOmp.step (the
    all-cores CSR variant) against qo.cheby: three operators, dt = +0.7 and -0.7, 3 steps each, < 1e-13 after every step."""
    from oracle import ref_c
    for name, H, psi0, Delta, E_min in _c_port_cases():
        H = H.tocsr()
        H.sort_indices()
        Hc = H.tocsc()
        Hc.sort_indices()
        for dt in (0.7, -0.7):
            wrk = qo.ChebyWrk(psi0, Delta, E_min, abs(dt))
            a = wrk.coeffs[:wrk.n_coeffs].copy()
            ref, c_csc = psi0.copy(), psi0.copy()
            omp = ref_c.ChebyCsrOmp(H.indptr, H.indices, H.data, psi0)
            for k in range(3):
                qo.cheby(ref, H, dt, wrk)
                ref_c.cheby_csc(Hc.indptr, Hc.indices, Hc.data, c_csc, a, Delta, E_min, dt)
                omp.step(a, Delta, E_min, dt)
                assert np.linalg.norm(c_csc - ref) < 1e-13, (name, dt, k, "serial CSC")
                assert np.linalg.norm(omp.psi() - ref) < 1e-13, (name, dt, k, "OpenMP CSR")
            omp.close()
            one = psi0.copy()
            ref_c.cheby_csr_omp(H.indptr, H.indices, H.data, one, a, Delta, E_min, dt)
            single = qo.cheby(psi0.copy(), H, dt, qo.ChebyWrk(psi0, Delta, E_min, abs(dt)))
            assert np.linalg.norm(one - single) < 1e-13, (name, dt, "one-shot OpenMP CSR")


def test_c_port_matches_50_digit_arithmetic():
    """The C port against exp(-i H dt) psi in 50-digit arithmetic (mpmath) at N = 24: what test/test_cheby.jl:24-47 compares
    with, independent of NumPy and of the double-precision oracle."""
    pytest.importorskip("mpmath")
    from oracle import qp_oracle_mp as qmp
    from oracle import ref_c
    rng = np.random.default_rng(5)
    N = 24
    X = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    Hd = (X + X.conj().T) / 2
    Hd[np.abs(Hd) < 0.6] = 0.0            # sparse, still Hermitian (|.| is symmetric)
    H = sp.csr_matrix(Hd)
    ev = np.linalg.eigvalsh(Hd)
    Delta, E_min = float(1.02 * (ev[-1] - ev[0])), float(ev[0] - 0.01 * (ev[-1] - ev[0]))
    psi0 = _rand_state(N, rng)
    for dt in (0.4, -0.4):
        a = qo.cheby_coeffs(Delta, abs(dt))
        Hc = H.tocsc()
        got = ref_c.cheby_csc(Hc.indptr, Hc.indices, Hc.data, psi0.copy(), a, Delta, E_min, dt)
        exact = np.array([complex(z) for z in qmp.expm_apply(Hd, psi0, dt)])
        assert np.linalg.norm(got - exact) < 1e-12, dt


def test_newton_c_port_matches_the_numpy_oracle():
    """ref_c.newton_csc (oracle/newton_ref.c: serial CSC mat-vec, the reference's sequential modified Gram-Schmidt, its two
    combinations -- src/arnoldi.jl:79-96, src/newton.jl:346-367 -- with the NumPy oracle's own host algebra between them)
    against qo.newton and against the dense exponential: fixture F4 (N = 256, result, restart count and the first restart's
    Hessenberg matrix as stored), an N = 4096 Liouvillian over +dt, +dt, -dt with equal restart counts, and the reference's
    `func = exp` variant (test/test_newton.jl:166-175).  The bench's C3 `cpu_baseline` rests on this."""
    import os
    import scipy.linalg as sla
    from oracle import ref_c
    import qprop_amd.synth as synth
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "F4_newton_liouvillian_n256.npz"))
    Lm = sp.csr_matrix((f["vals"], f["col"], f["rowptr"]), shape=(int(f["n"]), int(f["n"])))
    Lc = Lm.tocsc()
    Lc.sort_indices()
    wc = ref_c.NewtonCscWrk(Lm.shape[0], m_max=int(f["m_max"]))
    rho = ref_c.newton_csc(Lc.indptr, Lc.indices, Lc.data, f["rho0"].copy(), float(f["dt"]), wc)
    assert np.linalg.norm(rho - f["result"]) < 1e-13 and wc.restarts == int(f["restarts"]) and wc.n_a == int(f["n_a"])
    assert abs(wc.radius - float(f["radius"])) < 1e-12 * float(f["radius"])
    m_eff, Hess, nmv = ref_c.arnoldi_csc(Lc.indptr, Lc.indices, Lc.data, np.zeros((21, Lm.shape[0]), dtype=np.complex128), 20,
                                         f["rho0"] / np.linalg.norm(f["rho0"]), float(f["dt"]), True, 1e-14)
    assert m_eff == 20 and nmv == 20 and np.max(np.abs(Hess - f["first_Hess"])) < 1e-13
    # a non-extended sweep leaves the last sub-diagonal entry alone and does not normalise the last vector (src/arnoldi.jl:88)
    q = np.zeros((6, Lm.shape[0]), dtype=np.complex128)
    m5, H5, _ = ref_c.arnoldi_csc(Lc.indptr, Lc.indices, Lc.data, q, 5, f["rho0"] / np.linalg.norm(f["rho0"]), 1.0, False, 1e-15)
    H5o = np.zeros((6, 6), dtype=np.complex128)
    qo_ = [np.zeros(Lm.shape[0], dtype=np.complex128) for _ in range(6)]
    assert qo.arnoldi(H5o, qo_, 5, f["rho0"] / np.linalg.norm(f["rho0"]), Lm, 1.0, extended=False) == m5 == 5
    assert np.max(np.abs(H5 - H5o)) < 1e-13 and H5[5, 4] == 0 and np.linalg.norm(q[5] - qo_[5]) < 1e-12
    # breakdown: an eigenvector start gives m = 1 and the eigenstate shortcut (src/newton.jl:289-295)
    D = sp.diags(np.arange(1.0, 9.0)).tocsc().astype(np.complex128)
    e3 = np.zeros(8, dtype=np.complex128)
    e3[3] = 1.0
    w8 = ref_c.NewtonCscWrk(8, m_max=5)
    out = ref_c.newton_csc(D.indptr, D.indices, D.data, e3.copy(), 0.3, w8)
    assert w8.restarts == 0 and w8.n_matvec == 1 and abs(out[3] - np.exp(-1j * 4.0 * 0.3)) < 1e-14
    # (for a state of norm beta != 1 the reference's shortcut evaluates func(beta * Hess[1, 1]): kept, as the NumPy oracle keeps it)
    out2 = ref_c.newton_csc(D.indptr, D.indices, D.data, 2.0 * e3, 0.3, w8)
    assert np.linalg.norm(out2 - qo.newton(2.0 * e3, D.tocsr(), 0.3, qo.NewtonWrk(e3, m_max=5))) < 1e-14
    # N = 4096
    Lb = synth.liouvillian_tridiag(64)
    N = Lb.shape[0]
    Lbc = Lb.tocsc()
    Lbc.sort_indices()
    rho0 = synth.random_state(N)
    ref, got = rho0.copy(), rho0.copy()
    w, wc = qo.NewtonWrk(ref, m_max=20), ref_c.NewtonCscWrk(N, m_max=20)
    for dt in (0.5, 0.5, -0.3):
        qo.newton(ref, Lb, dt, w)
        ref_c.newton_csc(Lbc.indptr, Lbc.indices, Lbc.data, got, dt, wc)
        assert np.linalg.norm(got - ref) < 1e-12 and wc.restarts == w.restarts, dt
    # func = exp on a small dense system against the dense exponential (the reference's own check, 1e-10)
    rng = np.random.default_rng(11)
    A = sp.csc_matrix((rng.standard_normal((40, 40)) + 1j * rng.standard_normal((40, 40))) * (rng.random((40, 40)) < 0.3) * 0.4)
    A.sort_indices()
    x0 = _rand_state(40, rng)
    wA = ref_c.NewtonCscWrk(40, m_max=10)
    y = ref_c.newton_csc(A.indptr, A.indices, A.data, x0.copy(), 0.7, wA, func=np.exp)
    assert np.linalg.norm(y - sla.expm(A.toarray() * 0.7) @ x0) < 1e-10
