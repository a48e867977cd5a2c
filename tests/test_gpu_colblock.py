"""GPU parity tests of the column-blocked mirror (csrc/kernels_colblock.hip; include/qprop.h: qp_operator_colblock_info):
operators with IRREGULAR columns -- the reference's generators hold any sparse matrix (src/generators.jl:634-645), and
cheby! / arnoldi! / mul! are `mul!(v, H, Psi)` with it (src/cheby.jl:177, :191; src/arnoldi.jl:81).  Every case is compared
with the oracle AND with the same operator's ordinary row-block kernel (knob colblock switched off on the live operator).
Tolerance 1e-10 on |psi> (BASELINE north_star), 1e-13 relative between the two kernels (same products, another summation tree)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
from oracle import ref_c  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture()
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _rand_state(N, rng):
    psi = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    return psi / np.linalg.norm(psi)


def _random_sparse(nr, nc, per_row, rng, hermitian=False, real=False, empty_rows=()):
    """Columns drawn per row anywhere in [0, nc); complex (or real) values of modulus <= 1 / per_row... (spectrum in the unit disc)."""
    rows = np.repeat(np.arange(nr), per_row)
    cols = rng.integers(0, nc, size=nr * per_row)
    vals = (rng.standard_normal(nr * per_row) + (0 if real else 1j) * rng.standard_normal(nr * per_row)) / (3.0 * per_row)
    keep = ~np.isin(rows, np.asarray(empty_rows, dtype=np.int64))
    A = sp.coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=(nr, nc)).tocsr()
    if hermitian:
        A = (A + A.getH()).tocsr() * 0.5
    A.sum_duplicates()
    A.sort_indices()
    return A.astype(np.complex128)


def _forced(ctx, log2w=8):
    """knobs for small operators: the mirror for anything that fits its limits, 2^log2w columns per block"""
    ctx.tuning_set("colblock", 2)
    ctx.tuning_set("cb_log2w", log2w)


@pytest.mark.parametrize("shape", [(64, 64), (127, 127), (128, 300), (1000, 1000), (1025, 4099), (5000, 5000), (20000, 20000),
                                   (300, 64), (5681, 358), (40000, 70)])      # the last three: TALL operators (rows beyond the last column)
@pytest.mark.parametrize("real", [False, True])
def test_mul_through_the_mirror_matches_scipy_and_the_row_block_kernel(ctx, shape, real):
    """mul!(y, A, x, alpha, beta) (src/generators.jl:634-645): ragged last tile, rectangular operators (a rank's local rows
    of a partitioned H), empty rows, a real operator streaming its real copy."""
    nr, nc = shape
    rng = np.random.default_rng(nr * 7 + nc + real)
    _forced(ctx)
    A = _random_sparse(nr, nc, 9, rng, real=real, empty_rows=(0, nr // 2, nr - 1))
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, A)], 0, L.FMT_RBCSR)
    info = op.colblock_info()
    assert info["column_blocks"] == (nc + 255) // 256 * info["valid"] and info["rows_per_tile"] in (0, 64, 128)
    # (a tall, narrow operator can put more than 1024 entries into one (64-row, 256-column) cell: no mirror then, the
    # row-block kernel alone -- whose pad entries must still name an existing column: extended fuzzing, round 4)
    assert info["own_line_share"] > 0.5 or nc < 4096      # (few columns: the 64 gathers of a load share the handful of lines)
    x = _rand_state(nc, rng)
    y0 = _rand_state(nr, rng)
    xs, ys = L.State(ctx, data=x), L.State(ctx, data=y0)
    op.mul(xs, ys, 0.7 - 0.2j, -0.3 + 0.1j)
    got = ys.numpy()
    want = (0.7 - 0.2j) * (A @ x) + (-0.3 + 0.1j) * y0
    assert np.max(np.abs(got - want)) < 1e-13
    ctx.tuning_set("colblock", 0)        # the same operator through its row-block kernel
    ys.upload(y0)
    op.mul(xs, ys, 0.7 - 0.2j, -0.3 + 0.1j)
    assert np.max(np.abs(ys.numpy() - got)) < 1e-14


@pytest.mark.parametrize("N,per_row,log2w", [(3000, 8, 8), (4096, 16, 9), (10007, 5, 10), (40000, 16, 12)])
def test_cheby_step_matches_oracle_and_row_block_kernel(ctx, N, per_row, log2w):
    """cheby! (src/cheby.jl:150-213) on a Hermitian operator with random columns: both signs of dt, two steps."""
    rng = np.random.default_rng(N + per_row)
    _forced(ctx, log2w)
    H = _random_sparse(N, N, per_row, rng, hermitian=True)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, H)])
    assert op.format in (L.FMT_RBCSR, L.FMT_CSR) and op.colblock_info()["valid"] == 1     # irregular blocks are not Hermitian-packed
    psi0 = _rand_state(N, rng)
    for dt in (0.9, -0.9):
        wrk = L.ChebyWrk(ctx, N, 3.0, -1.5, abs(dt))
        owrk = qo.ChebyWrk(psi0, 3.0, -1.5, abs(dt))
        assert owrk.n_coeffs == wrk.n_coeffs
        want = psi0.copy()
        for _ in range(2):
            qo.cheby(want, H, dt, owrk)
        ctx.tuning_set("colblock", 2)
        ps = L.State(ctx, data=psi0)
        ctx.reset_stats()
        for _ in range(2):
            L.cheby(ps, op, dt, wrk)
        assert ctx.stats()["n_matvec"] == 2 * (wrk.n_coeffs - 1)
        got = ps.numpy()
        assert np.linalg.norm(got - want) < TOL
        ctx.tuning_set("colblock", 0)
        ps.upload(psi0)
        for _ in range(2):
            L.cheby(ps, op, dt, wrk)
        assert np.linalg.norm(ps.numpy() - got) < 1e-13
        # the normalisation check of the reference (src/cheby.jl:194-200) is a per-workgroup reduction of the row-block kernels:
        # a step that asks for it runs them, with the same result
        ctx.tuning_set("colblock", 2)
        ps.upload(psi0)
        for _ in range(2):
            L.cheby(ps, op, dt, wrk, check_normalization=True)
        assert np.linalg.norm(ps.numpy() - got) < 1e-13


def test_newton_and_arnoldi_through_the_mirror(ctx):
    """arnoldi! / newton! (src/arnoldi.jl:60-129, src/newton.jl:246-385) on a non-Hermitian operator with random columns: the
    columns' mat-vecs run through the mirror (the fused mat-vec + dots kernel is a row-block kernel and steps aside)."""
    N, m = 6000, 8
    rng = np.random.default_rng(5)
    _forced(ctx, 9)
    A = _random_sparse(N, N, 10, rng) - 0.05j * sp.identity(N)
    A = sp.csr_matrix(A)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, A)])
    assert op.colblock_info()["valid"] == 1
    psi0 = _rand_state(N, rng)
    owrk = qo.NewtonWrk(psi0, m_max=m)
    want = qo.newton(psi0.copy(), A, 0.5, owrk)
    wrk = L.NewtonWrk(ctx, N, m_max=m)
    ps = L.State(ctx, data=psi0)
    L.newton(ps, op, 0.5, wrk)
    got = ps.numpy()
    assert np.linalg.norm(got - want) < TOL
    ctx.tuning_set("colblock", 0)
    ps.upload(psi0)
    wrk2 = L.NewtonWrk(ctx, N, m_max=m)
    L.newton(ps, op, 0.5, wrk2)
    assert np.linalg.norm(ps.numpy() - got) < 1e-12 and wrk2.restarts == wrk.restarts


def test_time_dependent_generator_refreshes_the_mirror(ctx):
    """evaluate! (src/generators.jl: Operator(ops, coeffs)): the mirror's values follow qp_operator_set_coeffs -- drift plus two
    control terms, complex coefficients, a sparse (diagonal) control term through the partial update, all-real -> complex."""
    N = 5000
    rng = np.random.default_rng(11)
    _forced(ctx, 9)
    H0 = _random_sparse(N, N, 8, rng, real=True)
    H1 = _random_sparse(N, N, 3, rng, real=True)
    D = sp.diags(rng.standard_normal(N)).tocsr().astype(np.complex128)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, M) for M in (H0, H1, D)], 2, L.FMT_RBCSR)
    assert op.colblock_info()["valid"] == 1
    x = _rand_state(N, rng)
    xs, ys = L.State(ctx, data=x), L.State(ctx, n=N)
    for coeffs in ([1.0, 1.0], [0.3, -0.7], [0.3, 0.25], [0.5 + 0.5j, 0.1], [0.0, 0.0], [2.0, -1.0]):
        op.set_coeffs(coeffs)
        op.mul(xs, ys, 1.0, 0.0)
        want = H0 @ x + coeffs[0] * (H1 @ x) + coeffs[1] * (D @ x)
        assert np.max(np.abs(ys.numpy() - want)) < 1e-13, coeffs
    assert op.evaluate_info()["first_sparse_term"] in (-1, 2)
    rp, col, vals = op.get_csr()
    U = sp.csr_matrix((vals, col, rp), shape=(N, N))
    assert abs(U - (H0 + 2.0 * H1 - D)).max() < 1e-15


def test_auto_decision(ctx):
    """colblock = 1 (the default): the mirror is built for irregular gathers on a vector that outgrows the L2 -- not for a band,
    not for a lattice (Hermitian-packed and walked), not for a small operator, not for a dense one."""
    assert ctx.tuning_get("colblock") == 1 and ctx.tuning_get("cb_log2w") == 0
    N = 1 << 20
    rp, col, vals = synth.random_columns_csr(N)
    op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
    info = op.colblock_info()
    assert op.format == L.FMT_RBCSR and info["valid"] == 1 and info["column_blocks"] == 16 and info["log2_block_columns"] == 16
    assert info["own_line_share"] > 0.9
    assert info["rows_per_tile"] == 128 and info["longest_segment"] <= 1024
    # one cheby! step against the C oracle (serial CSC mat-vec), and against the row-block kernel
    psi0 = synth.random_state(N)
    wrk = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
    ps = L.State(ctx, data=psi0)
    L.cheby(ps, op, 1.0, wrk)
    got = ps.numpy()
    want = psi0.copy()      # (Hermitian: the CSC arrays of H are the CSR arrays of conj(H))
    ref_c.cheby_csc(rp, col.astype(np.int64), np.conj(vals), want, wrk.coeffs, 20.0, -10.0, 1.0)
    assert np.linalg.norm(got - want) < TOL
    ctx.tuning_set("colblock", 0)
    ps.upload(psi0)
    L.cheby(ps, op, 1.0, wrk)
    assert np.linalg.norm(ps.numpy() - got) < 1e-13
    ctx.tuning_set("colblock", 1)
    op.close()
    # a band: no mirror
    rpb, colb, valsb = synth.hermitian_offsets_csr(N)
    opb = L.Operator(ctx, [L.Matrix(ctx, N, N, rpb, colb, valsb)], 0, L.FMT_RBCSR)
    ib = opb.colblock_info()
    assert ib["valid"] == 0 and 0.0 < ib["own_line_share"] < 0.3
    opb.close()
    # columns drawn per row but NEAR the row (inside 4096-row windows): every gather is its own line, yet the lines stay in the
    # XCD's L2 as the rows stream by -- no mirror (101 us per term on the row-block kernel, 186 through a mirror)
    rw, cw, vw = synth.random_columns_csr(N, window=4096)
    opw = L.Operator(ctx, [L.Matrix(ctx, N, N, rw, cw, vw)])
    iw = opw.colblock_info()
    assert iw["valid"] == 0 and iw["own_line_share"] > 0.5
    opw.close()
    # small: no mirror (the vector fits the L2)
    n = 1 << 16
    r2, c2, v2 = synth.random_columns_csr(n)
    ops = L.Operator(ctx, [L.Matrix(ctx, n, n, r2, c2, v2)])
    assert ops.colblock_info()["valid"] == 0
    ctx.tuning_set("colblock", 0)
    assert L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)]).colblock_info()["valid"] == 0
