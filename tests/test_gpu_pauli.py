"""GPU parity tests of the qubit-register operator applied from its Pauli strings (csrc/engine_pauli.hip, include/qprop.h:
qp_pauli_operator_create): the reference would be handed the SAME generator as sparse matrices (src/generators.jl:634-645 mul!), so
the oracle runs on the matrix built from explicit Kronecker products of the Pauli matrices (synth.pauli_sum_matrix -- independent of the
mask arithmetic under test): cheby!, newton!, mul!, dot, lazy sums with time-dependent coefficients, the propagator interface."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.propagator as P  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture()
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _bound(strings):
    return float(sum(abs(complex(a)) for a, _ in strings))


def _random_strings(n, count, rng, labels="IXYZ"):
    out = []
    for _ in range(count):
        lab = "".join(rng.choice(list(labels), size=n, p=[0.55, 0.15, 0.15, 0.15] if labels == "IXYZ" else None))
        if set(lab) == {"I"}:
            lab = "Z" + lab[1:]
        out.append((float(rng.uniform(-1.0, 1.0)), L.pauli_masks(lab)))
    return out


@pytest.mark.parametrize("name,n", [("tfim", 12), ("xxz", 11), ("random", 10), ("random_y", 13)])
def test_pauli_cheby_matches_oracle_and_stored_matrix(ctx, name, n):
    """cheby! forward, forward, backward through the fused Pauli term against the oracle on the Kronecker-built matrix (1e-10) and
    against the library's own stored-matrix operator of the same generator (1e-12: another summation order)."""
    rng = np.random.default_rng(7 + n)
    strings = {"tfim": lambda: synth.tfim_pauli_terms(n), "xxz": lambda: synth.xxz_pauli_terms(n),
               "random": lambda: _random_strings(n, 14, rng), "random_y": lambda: _random_strings(n, 40, rng, labels="IYXZ")}[name]()
    N = 1 << n
    H = synth.pauli_sum_matrix(n, strings)
    assert abs(H - H.conj().T).max() < 1e-14
    b = 1.05 * _bound(strings)
    op = L.PauliOperator(ctx, n, [strings])
    assert op.format == L.FMT_MATFREE and op.nrows == N
    stored = L.Operator(ctx, [L.Matrix.from_scipy(ctx, H)])
    psi0 = synth.random_state(N, seed=n)
    dt = 10.0 / b
    wrk = L.ChebyWrk(ctx, N, 2 * b, -b, dt)
    psi, phi = L.State(ctx, data=psi0), L.State(ctx, data=psi0)
    ref = psi0.copy()
    owrk = qo.ChebyWrk(psi0, 2 * b, -b, dt)
    for sg in (1, 1, -1):
        ctx.reset_stats()
        L.cheby(psi, op, sg * dt, wrk)
        assert ctx.stats()["n_kernel_launches"] <= wrk.n_coeffs      # ONE launch per term (+ the result's copy): the fused Pauli term
        L.cheby(phi, stored, sg * dt, wrk)
        qo.cheby(ref, H, sg * dt, owrk)
        assert np.linalg.norm(psi.numpy() - ref) < TOL, sg
        assert np.linalg.norm(psi.numpy() - phi.numpy()) < 1e-12, sg
    assert abs(np.linalg.norm(psi.numpy()) - 1.0) < 1e-12


def test_pauli_mul_dot_and_scale(ctx):
    """mul!(y, A, x, alpha, beta) (src/generators.jl:634-645), dot(x, A, y), ScaledOperator: against NumPy on the Kronecker-built matrix."""
    n = 9
    rng = np.random.default_rng(3)
    strings = _random_strings(n, 25, rng)
    N = 1 << n
    H = synth.pauli_sum_matrix(n, strings)
    op = L.PauliOperator(ctx, n, [strings])
    x0, y0 = synth.random_state(N, seed=1), synth.random_state(N, seed=2)
    x, y = L.State(ctx, data=x0), L.State(ctx, data=y0)
    op.mul(x, y, alpha=0.7 - 0.2j, beta=-0.4 + 1.1j)
    want = (-0.4 + 1.1j) * y0 + (0.7 - 0.2j) * (H @ x0)
    assert np.linalg.norm(y.numpy() - want) < 1e-13
    op.mul(x, y)                                      # beta = 0: y is not read
    assert np.linalg.norm(y.numpy() - H @ x0) < 1e-13
    assert abs(op.dot(x, y) - np.vdot(x0, H @ (H @ x0))) < 1e-12
    op.set_scale(-2.5j)
    op.mul(x, y)
    assert np.linalg.norm(y.numpy() - (-2.5j) * (H @ x0)) < 1e-13


def test_pauli_time_dependent_lazy_sum_and_newton(ctx):
    """A lazy sum drift + two controlled terms (evaluate! rewrites the strings' coefficients: src/generators.jl:757-766), through
    cheby! with changing coefficients and through newton! (the matrix-free Arnoldi path), against the oracle."""
    n = 10
    N = 1 << n
    zz = [(-1.0, (0, (1 << i) | (1 << (i + 1)))) for i in range(n - 1)]
    xs = [(-1.0, (1 << i, 0)) for i in range(n)]
    ys = [(0.5, ((1 << i) | (1 << (i + 2)), (1 << i) | (1 << (i + 2)))) for i in range(n - 2)]       # Y_i Y_{i+2}
    op = L.PauliOperator(ctx, n, [zz, xs, ys], ncoeffs=2)
    Hz, Hx, Hy = (synth.pauli_sum_matrix(n, s_) for s_ in (zz, xs, ys))
    b = 1.05 * (_bound(zz) + 1.3 * _bound(xs) + 0.9 * _bound(ys))
    psi0 = synth.random_state(N, seed=5)
    wrk = L.ChebyWrk(ctx, N, 2 * b, -b, 0.2)
    psi = L.State(ctx, data=psi0)
    ref = psi0.copy()
    owrk = qo.ChebyWrk(psi0, 2 * b, -b, 0.2)
    for c1, c2 in ((1.0, 0.0), (0.3, 0.9), (-1.3, -0.5)):
        op.set_coeffs([c1, c2])
        L.cheby(psi, op, 0.2, wrk)
        qo.cheby(ref, (Hz + c1 * Hx + c2 * Hy).tocsr(), 0.2, owrk)
        assert np.linalg.norm(psi.numpy() - ref) < TOL, (c1, c2)
    nw = L.NewtonWrk(ctx, N, m_max=12)
    onw = qo.NewtonWrk(ref, m_max=12)
    H = (Hz - 1.3 * Hx - 0.5 * Hy).tocsr()
    for dt in (0.3, -0.3, 0.15):
        L.newton(psi, op, dt, nw)
        qo.newton(ref, H, dt, onw)
        assert np.linalg.norm(psi.numpy() - ref) < TOL and nw.restarts == onw.restarts, dt


def test_pauli_propagator_interface(ctx):
    """`hamiltonian(PauliSum, (PauliSum, eps))` through init_prop / prop_step / propagate (method cheby with the Arnoldi spectral range,
    and newton) against the same generator given as sparse matrices -- what a user of the reference would switch from."""
    n = 8
    zz = P.PauliSum(n, [(-1.0, {i: "Z", i + 1: "Z"}) for i in range(n - 1)]) + P.PauliSum(n, [(-0.1, {i: "Z"}) for i in range(n)])
    xs = P.PauliSum(n, [(-1.0, {i: "X"}) for i in range(n)])
    eps = lambda t: 0.8 * np.sin(3.0 * t)      # noqa: E731
    tlist = np.linspace(0.0, 1.0, 21)
    psi0 = synth.random_state(1 << n, seed=11)
    for method in ("cheby", "newton"):
        got = P.propagate(psi0, P.hamiltonian(zz, (xs, eps)), tlist, method=method)
        want = P.propagate(psi0, P.hamiltonian(zz.tocsr(), (xs.tocsr(), eps)), tlist, method=method)
        assert np.linalg.norm(np.asarray(got) - np.asarray(want)) < TOL, method
    p = P.init_prop(psi0, P.hamiltonian(zz, (xs, eps)), tlist, "cheby")
    assert p._dgen.op.format == L.FMT_MATFREE
    from qprop_amd.interfaces import check_propagator
    assert check_propagator(p, quiet=True)


def test_pauli_fast_paths_follow_the_coefficients(ctx):
    """The term kernel has a two-FMA path for X strings with real coefficients (a transverse field) and look-up-free paths for
    groups of one and of two strings (csrc/engine_pauli.hip: GS, FAST); which one runs is decided per launch from the CURRENT
    coefficients: a transverse-field chain with a real scale, then a complex one (Y-like phases on every string), then real again; an
    XX + YY chain (two strings per group, Z masks on the YY strings); every mul! against the Kronecker-built matrix."""
    n = 9
    N = 1 << n
    x0 = synth.random_state(N, seed=21)
    for strings in (synth.tfim_pauli_terms(n), synth.xxz_pauli_terms(n)):
        H = synth.pauli_sum_matrix(n, strings)
        op = L.PauliOperator(ctx, n, [strings])
        x, y = L.State(ctx, data=x0), L.State(ctx, data=x0)
        for scale in (1.0, -0.3 + 2.0j, 0.5, 1.0j, -2.0):
            op.set_scale(scale)
            op.mul(x, y)
            assert np.linalg.norm(y.numpy() - scale * (H @ x0)) < 1e-12, scale
        op.set_scale(1.0)
        b = 1.05 * _bound(strings)
        wrk = L.ChebyWrk(ctx, N, 2 * b, -b, 3.0 / b)
        psi = L.State(ctx, data=x0)
        L.cheby(psi, op, 3.0 / b, wrk)
        ref = qo.cheby(x0.copy(), H, 3.0 / b, qo.ChebyWrk(x0, 2 * b, -b, 3.0 / b))
        assert np.linalg.norm(psi.numpy() - ref) < TOL


def test_pauli_create_argument_errors(ctx):
    with pytest.raises(L.QPError):
        L.PauliOperator(ctx, 5, [[(1.0, (1, 0))]])                 # fewer than 64 rows
    with pytest.raises(L.QPError):
        L.PauliOperator(ctx, 8, [[(1.0, (1 << 8, 0))]])            # a qubit beyond the register
    with pytest.raises(L.QPError):
        L.PauliOperator(ctx, 8, [[(1.0, (1, 0))]], ncoeffs=2)      # more coefficients than terms
    op = L.PauliOperator(ctx, 8, [[(1.0, "IIIIIIXZ")]])
    with pytest.raises(L.QPError):
        op.get_csr()                                               # a matrix-free operator has no stored entries
