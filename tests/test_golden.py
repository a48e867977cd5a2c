"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU part: the oracle still reproduces the committed numbers (the parity target cannot drift
silently), the exact solutions agree with them, and the C ABI's host-side index work /
numerics reproduce the index fixtures bit for bit.  GPU part: the HIP path against the
committed vectors (tolerance ||delta psi|| < 1e-10, index work bit-exact)."""
import os
import sys

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
TOL = 1e-10


def _load(name):
    return np.load(os.path.join(G, name))


def _csr(f):
    n = int(f["n"])
    return sp.csr_matrix((f["vals"], f["col"], f["rowptr"]), shape=(n, n))


# ------------------------------------------------------------------ CPU: oracle vs fixtures

def test_f1_optomech_oracle_and_exact():
    f = _load("F1_optomech.npz")
    H = _csr(f)
    tlist = f["tlist"]
    _, st_c = qo.propagate(f["psi0"], H, tlist, "cheby", storage=True, E_min=float(f["E_min"]), E_max=float(f["E_max"]))
    _, st_n = qo.propagate(f["psi0"], H, tlist, "newton", storage=True, m_max=10)
    assert np.max(np.linalg.norm(st_c - f["cheby_states"], axis=0)) < 1e-13
    assert np.max(np.linalg.norm(st_n - f["newton_states"], axis=0)) < 1e-13
    exact = sla.expm(-1j * H.toarray() * tlist[-1]) @ f["psi0"]
    assert np.linalg.norm(f["cheby_states"][:, -1] - exact) < TOL
    assert np.linalg.norm(f["newton_states"][:, -1] - exact) < TOL


def test_f3_f4_f5_oracle_reproduces():
    f = _load("F3_cheby_c2_n256.npz")
    H = _csr(f)
    wrk = qo.ChebyWrk(f["psi0"], float(f["Delta"]), float(f["E_min"]), 1.0)
    assert np.array_equal(wrk.coeffs, f["coeffs"]) and len(wrk.coeffs) == 32
    psi = f["psi0"].copy()
    for k, dt in enumerate(f["dts"]):
        qo.cheby(psi, H, float(dt), wrk)
        assert np.linalg.norm(psi - f["states"][:, k + 1]) < 1e-13
    exact = sla.expm(-1j * H.toarray() * 2.0) @ f["psi0"]          # +1 +1 +1 -1
    assert np.linalg.norm(f["states"][:, -1] - exact) < TOL
    f = _load("F4_newton_liouvillian_n256.npz")
    Lm = _csr(f)
    nw = qo.NewtonWrk(f["rho0"], m_max=int(f["m_max"]))
    rho = qo.newton(f["rho0"].copy(), Lm, float(f["dt"]), nw, record=True)
    assert np.linalg.norm(rho - f["result"]) < 1e-13 and nw.restarts == int(f["restarts"])
    assert np.linalg.norm(f["result"] - sla.expm(-1j * Lm.toarray() * float(f["dt"])) @ f["rho0"]) < TOL
    f = _load("F5_specrange_n300.npz")
    Hh = _csr(f)
    ritz = qo.ritzvals(Hh, f["state"], 20, 60, prec=1e-3)
    assert len(ritz) == len(f["ritz"]) and np.max(np.abs(ritz - f["ritz"])) < 1e-10
    ev = np.linalg.eigvalsh(Hh.toarray())
    assert float(f["E_min"]) <= ev[0] and ev[-1] <= float(f["E_max"])


def test_f4_host_numerics_via_cabi():
    """The C ABI's Hessenberg eigenvalues / Leja ordering / Newton coefficients reproduce the
    first-restart intermediates of F4 from the stored Hessenberg matrix."""
    f = _load("F4_newton_liouvillian_n256.npz")
    m = 20
    Hess = np.asfortranarray(f["first_Hess"])
    ritz = L.hessenberg_eigvals(Hess, m, accumulate=True)
    off = 0
    for j in range(1, m + 1):
        assert np.max(np.abs(np.sort_complex(ritz[off:off + j]) - np.sort_complex(f["first_ritz"][off:off + j]))) < 1e-11
        off += j
    radius = 1.2 * np.max(np.abs(f["first_ritz"]))
    assert abs(radius - float(f["radius"])) < 1e-12 * radius
    leja, n = L.extend_leja(np.zeros(m, dtype=complex), 0, f["first_ritz"].copy(), m)
    assert n == m and np.array_equal(leja[:m], f["leja"][:m])          # selection order: bit-exact
    a, n_a = L.extend_newton_coeffs(np.zeros(m, dtype=complex), 0, f["leja"][:m], "expmi", m, float(f["radius"]))
    # divided differences over 20 points in one sweep lose ~7 digits in the last coefficients
    # (inherent cancellation; libm exp vs NumPy exp is enough to show it).  The propagated
    # state is insensitive to it (test_gpu_f4_newton holds 1e-10).
    assert np.max(np.abs(a[:8] - f["a"][:8])) < 1e-12 and np.max(np.abs(a[:m] - f["a"][:m])) < 5e-9


def test_f6_index_work_via_cabi():
    f = _load("F6_index_work.npz")
    rowptr, col, vals = L.csc_to_csr(97, 97, f["colptr"], f["rowval"], f["nzval"], index_base=1)
    assert np.array_equal(rowptr, f["rowptr"]) and np.array_equal(col, f["col"]) and np.array_equal(vals, f["vals"])
    for parts in (3, 8):
        assert np.array_equal(L.partition_rows(f["rp1000"], parts, "rows"), f[f"parts_rows_{parts}"])
        assert np.array_equal(L.partition_rows(f["rp1000"], parts, "nnz"), f[f"parts_nnz_{parts}"])


def test_f2_c1_dense_oracle_and_exact():
    f = _load("F2_cheby_c1_dense128.npz")
    H, dt = f["H"], float(f["dt"])
    wrk = qo.ChebyWrk(f["psi0"], float(f["E_max"]) - float(f["E_min"]), float(f["E_min"]), dt)
    assert wrk.n_coeffs == int(f["n_coeffs"])
    psi = f["psi0"].copy()
    for k in range(200):
        qo.cheby(psi, H, dt, wrk)
        if (k + 1) % 50 == 0:
            assert np.linalg.norm(psi - f["checkpoints"][:, (k + 1) // 50 - 1]) < 1e-12
    ev, V = np.linalg.eigh(H)
    exact = V @ (np.exp(-1j * ev * 200 * dt) * (V.conj().T @ f["psi0"]))
    assert np.linalg.norm(f["checkpoints"][:, -1] - exact) < TOL


def test_f7_liouvillian_formulas():
    """The frozen superoperator equals the Lindblad map written out on rho (:TDSE: i d rho/dt = L rho)."""
    f = _load("F7_liouvillian_n6.npz")
    rho, n = f["rho"], f["rho"].shape[0]
    H = f["H0"] + float(f["eps"]) * f["H1"]
    comm = H @ rho - rho @ H
    diss = np.zeros_like(rho)
    for A in f["c_ops"]:
        G = A.conj().T @ A
        diss = diss + A @ rho @ A.conj().T - 0.5 * (G @ rho + rho @ G)
    vec = lambda M: np.ascontiguousarray(M.T).reshape(-1)       # noqa: E731
    assert np.linalg.norm(f["Lrho_TDSE"] - vec(comm + 1j * diss)) < 1e-12
    assert np.linalg.norm(f["Lrho_LvN"] - vec(1j * comm + diss)) < 1e-12
    assert f["L_TDSE"].shape == (n * n, n * n)


# ------------------------------------------------------------------ GPU: HIP path vs fixtures

@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_f1_optomech(ctx):
    import qprop_amd.propagator as P
    f = _load("F1_optomech.npz")
    H = _csr(f)
    _, st = P.propagate(f["psi0"], (H,), f["tlist"], method="cheby", storage=True, ctx=ctx, E_min=float(f["E_min"]),
                        E_max=float(f["E_max"]))
    assert np.max(np.linalg.norm(st - f["cheby_states"], axis=0)) < TOL
    _, st = P.propagate(f["psi0"], (H,), f["tlist"], method="newton", storage=True, ctx=ctx, m_max=10)
    assert np.max(np.linalg.norm(st - f["newton_states"], axis=0)) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [L.FMT_HRB, L.FMT_RBCSR, L.FMT_CSR])
def test_gpu_f3_cheby(ctx, fmt):
    f = _load("F3_cheby_c2_n256.npz")
    n = int(f["n"])
    Op = L.Operator(ctx, [L.Matrix(ctx, n, n, f["rowptr"], f["col"], f["vals"])], 0, fmt)
    wrk = L.ChebyWrk(ctx, n, float(f["Delta"]), float(f["E_min"]), 1.0)
    assert np.max(np.abs(wrk.coeffs - f["coeffs"])) < 5e-14 and wrk.n_coeffs == 32
    psi = L.State(ctx, data=f["psi0"])
    for k, dt in enumerate(f["dts"]):
        L.cheby(psi, Op, float(dt), wrk)
        assert np.linalg.norm(psi.numpy() - f["states"][:, k + 1]) < TOL
    rp, col, vals = Op.get_csr()      # index work: the device copy is the fixture, bit for bit
    assert np.array_equal(rp, f["rowptr"]) and np.array_equal(col, f["col"]) and np.array_equal(vals, f["vals"])


@pytest.mark.gpu
def test_gpu_f4_newton(ctx):
    f = _load("F4_newton_liouvillian_n256.npz")
    n = int(f["n"])
    Op = L.Operator(ctx, [L.Matrix(ctx, n, n, f["rowptr"], f["col"], f["vals"])])
    wrk = L.NewtonWrk(ctx, n, m_max=int(f["m_max"]))
    rho = L.State(ctx, data=f["rho0"])
    L.newton(rho, Op, float(f["dt"]), wrk)
    assert np.linalg.norm(rho.numpy() - f["result"]) < TOL
    assert abs(wrk.restarts - int(f["restarts"])) <= 1 and abs(wrk.radius - float(f["radius"])) < 1e-9
    # first-restart Hessenberg matrix from the device Arnoldi
    q = L.Krylov(ctx, n, 21)
    Hess = np.zeros((21, 21), dtype=complex, order="F")
    v = f["rho0"] / np.linalg.norm(f["rho0"])
    assert L.arnoldi(Hess, q, 20, L.State(ctx, data=v), Op, float(f["dt"])) == 20
    assert np.max(np.abs(Hess - f["first_Hess"])) < 1e-11


@pytest.mark.gpu
def test_gpu_f5_specrange(ctx):
    f = _load("F5_specrange_n300.npz")
    n = int(f["n"])
    Op = L.Operator(ctx, [L.Matrix(ctx, n, n, f["rowptr"], f["col"], f["vals"])])
    st = L.State(ctx, data=f["state"])
    ritz = L.ritzvals(Op, st, 20, 60, prec=1e-3)
    assert len(ritz) == len(f["ritz"]) and np.max(np.abs(ritz - f["ritz"])) < 1e-8
    lo, hi = L.specrange_arnoldi(Op, st, prec=1e-4)
    assert abs(lo - float(f["E_min"])) < 1e-8 and abs(hi - float(f["E_max"])) < 1e-8


@pytest.mark.gpu
def test_gpu_f2_c1_dense(ctx):
    """BASELINE config C1: N = 128 dense Hermitian, Cheby, 200 steps, through propagate()."""
    import qprop_amd.propagator as P
    f = _load("F2_cheby_c1_dense128.npz")
    dt = float(f["dt"])
    tlist = dt * np.arange(201)
    _, st = P.propagate(f["psi0"], (f["H"],), tlist, method="cheby", storage=True, ctx=ctx, E_min=float(f["E_min"]),
                        E_max=float(f["E_max"]), specrange_buffer=0.0)
    for k in range(4):
        assert np.linalg.norm(st[:, 50 * (k + 1)] - f["checkpoints"][:, k]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("convention", ["TDSE", "LvN"])
def test_gpu_f7_liouvillian(ctx, convention):
    """Sparse superoperator and matrix-free operator (both implementations) against the frozen L rho."""
    import qprop_amd.propagator as P
    f = _load("F7_liouvillian_n6.npz")
    vec = np.ascontiguousarray(f["rho"].T).reshape(-1)
    want = f[f"Lrho_{convention}"]
    gen = P.liouvillian((f["H0"], (f["H1"], lambda t: float(f["eps"]))), list(f["c_ops"]), convention=convention)
    Lsp = gen.ops[0] + float(f["eps"]) * gen.ops[1]
    assert np.max(np.abs(Lsp.toarray() - f[f"L_{convention}"])) < 1e-13
    x = L.State(ctx, data=vec)
    for fused in (4096, 0):
        L.tuning_set("liouville_fused_n", fused)
        try:
            Lmf = L.Liouvillian(ctx, [f["H0"], f["H1"]], list(f["c_ops"]), ncoeffs=1, convention=convention)
            Lmf.set_coeffs([float(f["eps"])])
            y = L.State(ctx, n=len(vec))
            Lmf.mul(x, y)
            assert np.linalg.norm(y.numpy() - want) < 1e-12
        finally:
            L.tuning_set("liouville_fused_n", 320)
    Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, sp.csr_matrix(Lsp))])
    y = L.State(ctx, n=len(vec))
    Op.mul(x, y)
    assert np.linalg.norm(y.numpy() - want) < 1e-12
