"""GPU parity tests of the dense operator format (QP_FMT_DENSE, csrc/kernels_dense.hip): the reference's own dense tests
restated on the device -- test/test_cheby.jl:24-47 (N = 1000 `Hermitian(rand(ComplexF64, N, N))`, 267 / 268 coefficients,
1e-10 against dense `exp`), test/test_newton.jl:7-66 (Hermitian, m_max = 5) and :70-125 (non-Hermitian, m_max = 50),
BASELINE configs[0] (fixture F2) -- the format choice, the row-sum kernel against the oracle's mat-vec for awkward shapes,
and the batched step (H [psi_1 .. psi_b] on the fp64 matrix cores) against the oracle state by state.
Tolerance 1e-10 on |psi> as everywhere (BASELINE north_star)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-10
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _rand_state(N, rng):
    psi = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    return psi / np.linalg.norm(psi)


def _julia_hermitian_rand(N, rng):
    """`Hermitian(rand(ComplexF64, N, N))`: the upper triangle of a uniform [0, 1)^2 matrix, real diagonal."""
    X = rng.random((N, N)) + 1j * rng.random((N, N))
    H = np.triu(X) + np.triu(X, 1).conj().T
    H[np.diag_indices(N)] = H[np.diag_indices(N)].real
    return H


def test_auto_takes_the_dense_format_above_the_density_threshold(ctx):
    rng = np.random.default_rng(1)
    N = 96
    H = synth.dense_hermitian(N, rho=3.0, rng=rng)
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
    assert op.format == L.FMT_DENSE and op.fill_info() == 0
    rp, col, vals = op.get_csr()
    assert np.array_equal(rp, np.arange(N + 1) * N) and np.array_equal(col, np.tile(np.arange(N), N))
    assert np.array_equal(vals.reshape(N, N), H)
    # 80 % of the positions stored: dense, the missing ones explicit zeros (shown by get_csr, counted by fill_info)
    mask = rng.random((N, N)) < 0.8
    A = sp.csr_matrix(np.where(mask, H, 0))
    A.eliminate_zeros()
    op80 = L.Operator(ctx, [L.Matrix.from_scipy(ctx, A)])
    assert op80.format == L.FMT_DENSE and op80.fill_info() == N * N - A.nnz
    assert np.array_equal(op80.get_csr()[2].reshape(N, N), A.toarray())
    # 50 %: stays sparse; a dense layout can still be asked for, and refused formats stay refused
    A50 = sp.csr_matrix(np.where(rng.random((N, N)) < 0.5, H, 0))
    A50.eliminate_zeros()
    assert L.Operator(ctx, [L.Matrix.from_scipy(ctx, A50)]).format != L.FMT_DENSE
    assert L.Operator(ctx, [L.Matrix.from_scipy(ctx, A50)], 0, L.FMT_DENSE).format == L.FMT_DENSE
    ctx.tuning_set("dense_auto", 0)
    try:
        assert L.Operator(ctx, [L.Matrix.from_dense(ctx, H)]).format == L.FMT_CSR
    finally:
        ctx.tuning_set("dense_auto", 1)
    with pytest.raises(L.QPError):
        L.Operator(ctx, [L.Matrix.from_dense(ctx, H)], 0, 6)


@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (63, 63), (64, 65), (65, 64), (257, 300), (1000, 1000), (1025, 1023), (300, 16384 + 7)])
@pytest.mark.parametrize("real", [False, True])
def test_dense_mul_matches_numpy(ctx, shape, real):
    """mul!(y, A, x, alpha, beta) (src/generators.jl:634-645) through the row-sum kernel, rectangular and ragged-tail shapes;
    an all-real operator streams its real copy."""
    nr, nc = shape
    rng = np.random.default_rng(nr * 7 + nc)
    A = rng.standard_normal((nr, nc)) + (0 if real else 1j) * rng.standard_normal((nr, nc))
    M = L.Matrix(ctx, nr, nc, np.arange(nr + 1, dtype=np.int64) * nc, np.tile(np.arange(nc, dtype=np.int32), nr),
                 A.astype(np.complex128).reshape(-1))
    op = L.Operator(ctx, [M])
    assert op.format == L.FMT_DENSE
    x, y0 = _rand_state(nc, rng), _rand_state(nr, rng)
    X = L.State(ctx, data=x)
    for alpha, beta in ((1, 0), (2, 1), (0.5 - 1j, 0.25j)):
        Y = L.State(ctx, data=y0)
        op.mul(X, Y, alpha, beta)
        ref = alpha * (A @ x) + beta * y0
        assert np.linalg.norm(Y.numpy() - ref) < 1e-12 * max(1.0, np.linalg.norm(ref))


def test_dense_lazy_sum_and_evaluate(ctx):
    """Operator = drift + two controls on dense terms (one of them sparse: the union is complete anyway), coefficient
    updates (evaluate!, src/generators.jl:757-766), scale, dot(x, A, y)."""
    rng = np.random.default_rng(5)
    N = 130
    H0 = synth.dense_hermitian(N, rho=2.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    H2 = sp.diags(np.linspace(-1, 1, N)).tocsr().astype(complex)
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H0), L.Matrix.from_dense(ctx, H1), L.Matrix.from_scipy(ctx, H2)], 2)
    assert op.format == L.FMT_DENSE
    x, y = _rand_state(N, rng), _rand_state(N, rng)
    X, Yv = L.State(ctx, data=x), L.State(ctx, data=y)
    for c in ([0.3, -1.2], [0.7 + 0.2j, 0.1j], [0.0, 0.0]):
        op.set_coeffs(c)
        ref = H0 + c[0] * H1 + c[1] * H2.toarray()
        out = L.State(ctx, n=N)
        op.mul(X, out)
        assert np.linalg.norm(out.numpy() - ref @ x) < 1e-12
        assert abs(op.dot(Yv, X) - np.vdot(y, ref @ x)) < 1e-12
    op.set_scale(0.5j)
    out = L.State(ctx, n=N)
    op.mul(X, out)
    assert np.linalg.norm(out.numpy() - 0.5j * (H0 @ x)) < 1e-12


def test_dense_cheby_reference_test_n1000(ctx):
    """test/test_cheby.jl:6-49 on the device: N = 1000 `Hermitian(rand(ComplexF64, N, N))`, dt = 0.5, 267 or 268
    coefficients, cheby! against exp(-i H dt) to 1e-10 -- and against the oracle on the same inputs."""
    rng = np.random.default_rng(1)
    N, dt = 1000, 0.5
    H = _julia_hermitian_rand(N, rng)
    psi0 = rng.random(N) + 1j * rng.random(N)
    psi0 /= np.linalg.norm(psi0)
    evals, V = np.linalg.eigh(H)
    expected = V @ (np.exp(-1j * evals * dt) * (V.conj().T @ psi0))
    E_min, Delta = evals[0], evals[-1] - evals[0]
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
    assert op.format == L.FMT_DENSE
    wrk = L.ChebyWrk(ctx, N, Delta, E_min, dt)
    assert wrk.n_coeffs in (267, 268)
    psi = L.State(ctx, data=psi0)
    ctx.reset_stats()
    L.cheby(psi, op, dt, wrk)
    assert ctx.stats()["n_matvec"] == wrk.n_coeffs - 1
    out = psi.numpy()
    assert np.linalg.norm(out - expected) < TOL
    owrk = qo.ChebyWrk(psi0, Delta, E_min, dt)
    assert owrk.n_coeffs == wrk.n_coeffs
    assert np.linalg.norm(out - qo.cheby(psi0.copy(), H, dt, owrk)) < TOL
    # backward undoes forward (src/cheby.jl:158-162, :211), with the normalisation check on (src/cheby.jl:194-200)
    L.cheby(psi, op, -dt, wrk, check_normalization=True)
    assert np.linalg.norm(psi.numpy() - psi0) < TOL
    # the same operator forced through the CSR kernels: the two device formats agree far inside the tolerance
    op_csr = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)], 0, L.FMT_CSR)
    psi2 = L.State(ctx, data=psi0)
    L.cheby(psi2, op_csr, dt, wrk)
    assert np.linalg.norm(psi2.numpy() - out) < 1e-12


@pytest.mark.parametrize("hermitian,m_max", [(True, 5), (False, 50)])
def test_dense_newton_reference_tests_n1000(ctx, hermitian, m_max):
    """test/test_newton.jl:7-66 (random Hermitian, spectral radius 10, m_max = 5) and :70-125 (random non-Hermitian,
    m_max = 50): newton! with max_restarts = 200 against exp(-i H dt) psi to 1e-10, and the oracle's restart count."""
    rng = np.random.default_rng(7 if hermitian else 8)
    N, dt = 1000, 0.5
    H = synth.dense_hermitian(N, rho=10.0, rng=rng) if hermitian else synth.dense_nonhermitian(N, rho=10.0, rng=rng)
    psi0 = _rand_state(N, rng)
    import scipy.linalg as sla
    expected = sla.expm(-1j * H * dt) @ psi0
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
    assert op.format == L.FMT_DENSE
    wrk = L.NewtonWrk(ctx, N, m_max=m_max)
    psi = L.State(ctx, data=psi0)
    L.newton(psi, op, dt, wrk, max_restarts=200)
    assert np.linalg.norm(psi.numpy() - expected) < TOL
    owrk = qo.NewtonWrk(psi0, m_max=m_max)
    ref = qo.newton(psi0.copy(), H, dt, owrk, max_restarts=200)
    assert np.linalg.norm(psi.numpy() - ref) < TOL and wrk.restarts == owrk.restarts


def test_dense_c1_fixture(ctx):
    """BASELINE configs[0] (fixture F2: N = 128 dense Hermitian, 200 Chebyshev steps, checkpoints every 50) step by step
    through the dense row-sum kernel."""
    f = np.load(os.path.join(GOLD, "F2_cheby_c1_dense128.npz"))
    H, psi0 = f["H"], f["psi0"]
    N = H.shape[0]
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
    assert op.format == L.FMT_DENSE
    wrk = L.ChebyWrk(ctx, N, float(f["E_max"]) - float(f["E_min"]), float(f["E_min"]), float(f["dt"]))
    assert wrk.n_coeffs == int(f["n_coeffs"])
    psi = L.State(ctx, data=psi0)
    k = 0
    for step in range(200):
        L.cheby(psi, op, float(f["dt"]), wrk)
        if (step + 1) % 50 == 0:
            assert np.linalg.norm(psi.numpy() - f["checkpoints"][:, k]) < TOL
            k += 1


@pytest.mark.parametrize("N,batch", [(33, 1), (33, 8), (128, 31), (128, 64), (1000, 64), (515, 40), (1000, 96), (2050, 40), (3203, 64)])
@pytest.mark.parametrize("real", [False, True])
def test_dense_batched_cheby_on_the_matrix_cores(ctx, N, batch, real):
    """qp_cheby_step_batched of a dense operator: H [psi_1 .. psi_b] as a dense panel contraction on the fp64 matrix cores
    (v_mfma_f64_16x16x4_f64), recurrence + accumulate in the tile's epilogue; every state against the oracle's cheby!,
    forward and backward; ragged N (tile edges, inner dimension not a multiple of 4) and panel widths that do not fill
    a tile.  The three tile shapes of the kernel are all taken: 16 x 16 (narrow panels, or too few rows for wider tiles to
    fill the chip), 16 x 32 (N = 2050, b = 40) and 32 x 32 (N = 3203, b = 64)."""
    rng = np.random.default_rng(N + batch)
    H = synth.dense_hermitian(N, rho=4.0, rng=rng)
    if real:
        H = H.real.astype(np.complex128)
    ev = np.linalg.eigvalsh(H)
    Delta, E_min, dt = ev[-1] - ev[0] + 0.5, ev[0] - 0.25, 0.4
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
    assert op.format == L.FMT_DENSE
    states = np.stack([_rand_state(N, rng) for _ in range(batch)], axis=1)
    panel = L.State(ctx, data=states.reshape(-1))
    wrk = L.ChebyWrk(ctx, N * batch, Delta, E_min, dt)
    ctx.reset_stats()
    L.cheby_batched(panel, op, dt, wrk, batch)
    st = ctx.stats()
    assert st["n_matvec"] == wrk.n_coeffs - 1 and st["n_kernel_launches"] <= wrk.n_coeffs     # one launch per term (+ a copy)
    got = panel.numpy().reshape(N, batch)
    checked = range(batch) if N <= 1000 else sorted({0, 15, 16, 17, 31, 32, batch - 1} & set(range(batch)))   # (the oracle is O(N^2) per state and term)
    for s in checked:
        ref = qo.cheby(states[:, s].copy(), H, dt, qo.ChebyWrk(states[:, s], Delta, E_min, dt))
        assert np.linalg.norm(got[:, s] - ref) < TOL, s
    # the sparse panel kernels on the same operator (knob dense_panel_mfma 0): same states to 1e-12
    ctx.tuning_set("dense_panel_mfma", 0)
    try:
        panel2 = L.State(ctx, data=states.reshape(-1))
        L.cheby_batched(panel2, op, dt, wrk, batch)
        assert np.linalg.norm(panel2.numpy() - panel.numpy()) < 1e-12 * np.sqrt(batch)
    finally:
        ctx.tuning_set("dense_panel_mfma", 1)
    L.cheby_batched(panel, op, -dt, wrk, batch)
    assert np.linalg.norm(panel.numpy().reshape(N, batch) - states) < TOL * np.sqrt(batch)


def test_dense_batched_full_size_properties(ctx):
    """The measured point (tools/point.py dense --batch 64): N = 4096, 64 states.  Norms, forward / backward round trip, linearity,
    and eight states against the single-state dense step."""
    rng = np.random.default_rng(4096)
    N, batch = 4096, 64
    H = synth.dense_hermitian(N, rho=10.0, rng=rng)
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
    assert op.format == L.FMT_DENSE
    Delta, E_min, dt = 24.0, -12.0, 0.25
    states = np.stack([_rand_state(N, rng) for _ in range(batch)], axis=1)
    panel = L.State(ctx, data=states.reshape(-1))
    wrk = L.ChebyWrk(ctx, N * batch, Delta, E_min, dt)
    L.cheby_batched(panel, op, dt, wrk, batch)
    got = panel.numpy().reshape(N, batch)
    assert np.max(np.abs(np.linalg.norm(got, axis=0) - 1.0)) < 1e-11
    wrk1 = L.ChebyWrk(ctx, N, Delta, E_min, dt)
    for s in range(0, batch, 8):
        one = L.State(ctx, data=states[:, s])
        L.cheby(one, op, dt, wrk1)
        assert np.linalg.norm(one.numpy() - got[:, s]) < 1e-12
    # linearity: the step of a combination of two states is the combination of their steps
    comb = (0.6 * states[:, 3] + 0.8j * states[:, 5])
    one = L.State(ctx, data=comb)
    L.cheby(one, op, dt, wrk1)
    assert np.linalg.norm(one.numpy() - (0.6 * got[:, 3] + 0.8j * got[:, 5])) < 1e-11
    L.cheby_batched(panel, op, -dt, wrk, batch)
    assert np.linalg.norm(panel.numpy().reshape(N, batch) - states) < TOL * np.sqrt(batch)


@pytest.mark.parametrize("N", [5, 8, 20, 32, 200, 16384 + 37])
def test_dense_check_normalization_partials_match_the_dense_grid(ctx, N):
    """the per-workgroup triples of check_normalization (src/cheby.jl:194-200) were sized with the sparse
    kernels' grid, not the dense row-sum kernel's -- too small for N <= 32 (lanes per row 8 / 16 / 32), four times too large
    from N = 16384 (four rows per wavefront).  A correct window must pass and leave the oracle's result; a window that does
    not hold the spectrum must raise, as the oracle does."""
    rng = np.random.default_rng(100 + N)
    if N < 1000:
        H = synth.dense_hermitian(N, rho=3.0, rng=rng)
        bound = 3.0
    else:       # (a large dense Hermitian matrix built cheaply: real symmetric low-rank-plus-diagonal, every entry non-zero)
        u = rng.standard_normal(N) / np.sqrt(N)
        H = np.outer(u, u).astype(np.complex128) + np.diag(rng.uniform(-1, 1, N))
        bound = 1.0 + float(u @ u)
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
    assert op.format == L.FMT_DENSE
    psi0 = _rand_state(N, rng)
    dt = 0.4
    wrk = L.ChebyWrk(ctx, N, 2.2 * bound, -1.1 * bound, dt)
    psi = L.State(ctx, data=psi0)
    for _ in range(3):      # (repeated: partials left over from an earlier step must not leak into the next reduction)
        L.cheby(psi, op, dt, wrk, check_normalization=True)
    ref = psi0.copy()
    owrk = qo.ChebyWrk(psi0, 2.2 * bound, -1.1 * bound, dt)
    for _ in range(3):
        qo.cheby(ref, H, dt, owrk, check_normalization=True)
    assert np.linalg.norm(psi.numpy() - ref) < TOL
    # one-sided violation: the whole spectrum lies below the window
    bad = L.ChebyWrk(ctx, N, 0.2 * bound, 2.0 * bound, dt)
    psi.upload(psi0)
    with pytest.raises(L.QPAssertionError, match="Incorrect normalization"):
        L.cheby(psi, op, dt, bad, check_normalization=True)
    with pytest.raises(AssertionError, match="Incorrect normalization"):
        qo.cheby(psi0.copy(), H, dt, qo.ChebyWrk(psi0, 0.2 * bound, 2.0 * bound, dt), check_normalization=True)
