"""GPU parity tests for QUBIT-REGISTER Hamiltonians (a Pauli string couples row and row XOR mask): transverse-field Ising and
XXZ chains (`synth.tfim_csr`, `synth.xxz_csr`).  64-row blocks map onto 64-row blocks, so the row blocks' column sections
take the block-map encoding (include/qprop.h: qp_operator_encoding_info) -- one byte of index per entry.  cheby! against the
oracle (src/cheby.jl:150-213), the device round trip of the pattern, every device format, knob block_map on / off."""
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


@pytest.fixture()
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _bound(model, n):
    return (1.0 * (n - 1) + 0.1 * n + 1.0 * n) if model == "tfim" else (1.5 * (n - 1) + 0.05 * n)


@pytest.mark.parametrize("model", ["tfim", "xxz"])
@pytest.mark.parametrize("n", [7, 10, 13])
@pytest.mark.parametrize("fmt", [L.FMT_AUTO, L.FMT_RBCSR, L.FMT_HRB, L.FMT_CSR])
def test_spin_chain_cheby_matches_oracle(ctx, model, n, fmt):
    N = 1 << n
    rp, col, vals = (synth.tfim_csr if model == "tfim" else synth.xxz_csr)(n)
    H = sp.csr_matrix((vals, col, rp), shape=(N, N))
    op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)], 0, fmt)
    r2, c2, v2 = op.get_csr()
    assert np.array_equal(r2, rp) and np.array_equal(c2, col) and np.array_equal(v2, vals)
    enc = op.encoding_info()
    if op.format == L.FMT_RBCSR and model == "tfim":      # (the XXZ chain's ragged rows, and the upper-only sections of the packed format, put different terms into a slot)
        nb = (N + 63) // 64
        # every block maps block to block: upper sections as block maps (the Ising chain's far entries alone would be stencil
        # blocks; its six in-block spin flips are not), none left with four bytes of index per entry
        assert enc["upper"]["block_map"] + enc["upper"]["stencil"] == nb and enc["upper"]["int32"] == 0
        if N >= 1 << 10:
            assert enc["upper"]["block_map"] > 0
    b = _bound(model, n)
    Delta, E_min, dt = 2.2 * b, -1.1 * b, 10.0 / b
    psi0 = synth.random_state(N, seed=n)
    wrk = L.ChebyWrk(ctx, N, Delta, E_min, dt)
    owrk = qo.ChebyWrk(psi0, Delta, E_min, dt)
    psi = L.State(ctx, data=psi0)
    ref = psi0.copy()
    for sgn in (1, 1, -1):
        L.cheby(psi, op, sgn * dt, wrk)
        qo.cheby(ref, H, sgn * dt, owrk)
    assert np.linalg.norm(psi.numpy() - ref) < TOL


def test_block_map_knob_changes_the_bytes_not_the_bits(ctx):
    """The same operator with the encoding switched off at creation: int32 columns (the far spin flips are more than 32767 rows
    away: no int16 distances), four times the index bytes, identical results."""
    n = 17
    N = 1 << n
    rp, col, vals = synth.tfim_csr(n)
    psi0 = synth.random_state(N)
    b = _bound("tfim", n)
    out, idx = [], []
    for knob in (1, 0):
        ctx.tuning_set("block_map", knob)
        op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)], 0, L.FMT_RBCSR)
        enc = op.encoding_info()["upper"]
        assert (enc["block_map"] > 0) == (knob == 1) and (enc["int32"] > 0) == (knob == 0)
        idx.append(op.layout_info()["index_bytes"])
        wrk = L.ChebyWrk(ctx, N, 2.2 * b, -1.1 * b, 10.0 / b)
        psi = L.State(ctx, data=psi0)
        for _ in range(2):
            L.cheby(psi, op, 10.0 / b, wrk)
        out.append(psi.numpy())
    ctx.tuning_set("block_map", 1)
    assert np.array_equal(out[0], out[1]) and idx[0] < 0.3 * idx[1]


def test_two_qubit_terms_time_dependent_and_newton(ctx):
    """A driven chain: drift (ZZ + Z) + a time-dependent transverse field as a second term (`evaluate!`), and a dissipative-like
    non-Hermitian variant through newton!."""
    n = 11
    N = 1 << n
    rp, col, vals = synth.tfim_csr(n, J=1.0, h=0.0, hz=0.2)            # diagonal only (h = 0 entries are explicit zeros)
    rx, cx, vx = synth.tfim_csr(n, J=0.0, h=1.0, hz=0.0)              # the transverse field alone
    H0 = sp.csr_matrix((vals, col, rp), shape=(N, N))
    H0.eliminate_zeros()
    HX = sp.csr_matrix((vx, cx, rx), shape=(N, N))
    HX.eliminate_zeros()
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, H0), L.Matrix.from_scipy(ctx, HX)], 1)
    psi0 = synth.random_state(N)
    b = _bound("tfim", n)
    wrk = L.ChebyWrk(ctx, N, 2.2 * b, -1.1 * b, 0.3)
    owrk = qo.ChebyWrk(psi0, 2.2 * b, -1.1 * b, 0.3)
    psi = L.State(ctx, data=psi0)
    ref = psi0.copy()
    for eps in (0.0, 0.4, -0.9, 1.0):
        op.set_coeffs([eps])
        L.cheby(psi, op, 0.3, wrk)
        qo.cheby(ref, sp.csr_matrix(H0 + eps * HX), 0.3, owrk)
    assert np.linalg.norm(psi.numpy() - ref) < TOL
    A = sp.csr_matrix(H0 + 0.7 * HX - 0.05j * sp.identity(N))
    opn = L.Operator(ctx, [L.Matrix.from_scipy(ctx, A)])
    assert opn.encoding_info()["upper"]["int32"] == 0
    nw = L.NewtonWrk(ctx, N, m_max=8)
    ps = L.State(ctx, data=psi0)
    L.newton(ps, opn, 0.05, nw)
    want = qo.newton(psi0.copy(), A, 0.05, qo.NewtonWrk(psi0, m_max=8))
    assert np.linalg.norm(ps.numpy() - want) < TOL


# ---- the value-dictionary mirror (include/qprop.h: qp_operator_value_encoding_info; csrc/kernels_coded.hip) ----------------

def test_value_dictionary_of_the_ising_chain(ctx):
    """A 17-spin Ising chain holds two couplings and a few dozen diagonal energies: every 64-row block fits a table, the mat-vec
    streams one byte per entry instead of eight, the reconstruction is exact and the results are bit-identical to the plain path."""
    n = 17
    N = 1 << n
    rp, col, vals = synth.tfim_csr(n)
    psi0 = synth.random_state(N)
    b = _bound("tfim", n)
    out = []
    for knob in (1, 0):
        ctx.tuning_set("value_dict", knob)
        op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
        info = op.value_encoding_info()
        assert info["valid"] == knob, info
        if knob:
            assert info["coded_bytes"] < 0.2 * info["plane_bytes"] and info["tables"] < (N // 64) and info["table_entries"] <= 66 * info["tables"]
        else:
            assert info["reason"] == "knob value_dict = 0"
        r2, c2, v2 = op.get_csr()
        assert np.array_equal(r2, rp) and np.array_equal(c2, col) and np.array_equal(v2, vals)
        wrk = L.ChebyWrk(ctx, N, 2.2 * b, -1.1 * b, 10.0 / b)
        psi = L.State(ctx, data=psi0)
        for _ in range(2):
            L.cheby(psi, op, 10.0 / b, wrk, check_normalization=(knob == 1))
        y = L.State(ctx, n=N)
        op.mul(psi, y, 0.3 - 0.2j, 0.0)
        out.append((psi.numpy(), y.numpy()))
    ctx.tuning_set("value_dict", 1)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    H = sp.csr_matrix((vals, col, rp), shape=(N, N))
    ref = psi0.copy()
    owrk = qo.ChebyWrk(psi0, 2.2 * b, -1.1 * b, 10.0 / b)
    for _ in range(2):
        qo.cheby(ref, H, 10.0 / b, owrk)
    assert np.linalg.norm(out[0][0] - ref) < TOL


def test_value_dictionary_follows_evaluate(ctx):
    """Two terms (drift ZZ + Z, control X) with per-step coefficients, complex ones included: the tables are recombined, not a
    value plane; every step against the oracle on the summed matrix, and bit for bit against the plain path."""
    n = 12
    N = 1 << n
    rp, col, vals = synth.tfim_csr(n, J=1.0, h=0.0, hz=0.2)
    rx, cx, vx = synth.tfim_csr(n, J=0.0, h=1.0, hz=0.0)
    H0 = sp.csr_matrix((vals, col, rp), shape=(N, N))
    H0.eliminate_zeros()
    HX = sp.csr_matrix((vx, cx, rx), shape=(N, N))
    HX.eliminate_zeros()
    psi0 = synth.random_state(N)
    b = _bound("tfim", n)
    res = []
    for knob in (1, 0):
        ctx.tuning_set("value_dict", knob)
        op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, H0), L.Matrix.from_scipy(ctx, HX)], 1, L.FMT_RBCSR)
        assert op.value_encoding_info()["valid"] == knob
        wrk = L.ChebyWrk(ctx, N, 2.2 * b, -1.1 * b, 0.3)
        psi = L.State(ctx, data=psi0)
        ref = psi0.copy()
        owrk = qo.ChebyWrk(psi0, 2.2 * b, -1.1 * b, 0.3)
        for eps in (0.0, 0.4, -0.9, 1.0):
            op.set_coeffs([eps])
            L.cheby(psi, op, 0.3, wrk)
            qo.cheby(ref, sp.csr_matrix(H0 + eps * HX), 0.3, owrk)
            assert np.linalg.norm(psi.numpy() - ref) < TOL
        # a complex coefficient (a non-Hermitian sum): mul! against the summed matrix
        op.set_coeffs([0.3 + 0.5j])
        y = L.State(ctx, n=N)
        op.mul(psi, y, 1.0, 0.0)
        want = (H0 + (0.3 + 0.5j) * HX) @ psi.numpy()
        assert np.linalg.norm(y.numpy() - want) < 1e-12 * max(1.0, np.linalg.norm(want))
        r2, c2, v2 = op.get_csr()
        S = sp.csr_matrix((v2, c2, r2), shape=(N, N))
        assert abs(S - (H0 + (0.3 + 0.5j) * HX)).max() < 1e-15
        res.append((psi.numpy(), y.numpy()))
    ctx.tuning_set("value_dict", 1)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("seed", range(12))
def test_value_dictionary_random_few_valued_operators(ctx, seed):
    """Fuzz: random patterns (ragged rows, empty rows, N below / not a multiple of 64, tall and wide shapes for mul!), values drawn
    from a small set (or too many of them: no mirror), one to three terms with random coefficients -- reconstruction exact,
    mul! and cheby! bit-identical to the plain path and within tolerance of the oracle."""
    rng = np.random.default_rng(9000 + seed)
    nrows = int(rng.choice([5, 63, 64, 65, 200, 1000, 4097]))
    square = seed % 3 != 2
    ncols = nrows if square else int(rng.choice([7, 358, 2 * nrows + 3]))
    nterms = int(rng.integers(1, 4))
    nvals = int(rng.choice([1, 3, 40, 100000]))
    palette = rng.standard_normal(nvals) + (1j * rng.standard_normal(nvals) if seed % 2 else 0.0)
    mats = []
    for _ in range(nterms):
        lens = rng.integers(0, 12, nrows)
        lens[rng.integers(0, nrows, max(1, nrows // 10))] = 0
        rows = np.repeat(np.arange(nrows), lens)
        cols = rng.integers(0, ncols, rows.size)
        v = palette[rng.integers(0, nvals, rows.size)]
        A = sp.coo_matrix((v, (rows, cols)), shape=(nrows, ncols)).tocsr()
        A.sum_duplicates()
        A.sort_indices()
        mats.append(A)
    ncoef = int(rng.integers(0, nterms + 1)) if nterms > 1 else int(rng.integers(0, 2))
    coefs = list(rng.standard_normal(ncoef) + 1j * rng.standard_normal(ncoef) * (seed % 2))
    x0 = rng.standard_normal(ncols) + 1j * rng.standard_normal(ncols)
    total = sum(((1.0 if l < nterms - ncoef else coefs[l - (nterms - ncoef)]) * mats[l] for l in range(nterms)), sp.csr_matrix((nrows, ncols)))
    outs = []
    for knob in (1, 0):
        ctx.tuning_set("value_dict", knob)
        op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, A) for A in mats], ncoef, L.FMT_RBCSR)
        info = op.value_encoding_info()
        if knob == 0:
            assert info["valid"] == 0
        elif nvals == 100000 and nrows >= 1000:
            assert info["valid"] == 0 and info["reason"] in ("a block with more than 256 distinct values", "no saving"), info
        if ncoef:
            op.set_coeffs(coefs)
        r2, c2, v2 = op.get_csr()
        S = sp.csr_matrix((v2, c2, r2), shape=(nrows, ncols))
        assert abs(S - total).max() < 1e-14 if S.nnz else True
        x = L.State(ctx, data=x0)
        y = L.State(ctx, n=nrows)
        op.mul(x, y, 0.7 + 0.1j, 0.0)
        assert np.linalg.norm(y.numpy() - (0.7 + 0.1j) * (total @ x0)) < 1e-12 * max(1.0, np.linalg.norm(total @ x0))
        rec = [info["valid"], y.numpy()]
        if square and ncoef == 0 or (square and all(abs(c.imag) == 0 for c in coefs)):
            Hh = (total + total.getH()).tocsr() * 0.5
            # (cheby! wants a Hermitian operator: run the Hermitian part through the same path as one more operator)
            oph = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Hh)], 0, L.FMT_RBCSR)
            bound = float(abs(Hh).sum(axis=1).max()) + 1e-3
            psi0 = x0 / np.linalg.norm(x0)
            wrk = L.ChebyWrk(ctx, nrows, 2.0 * bound, -bound, 0.2)
            psi = L.State(ctx, data=psi0)
            L.cheby(psi, oph, 0.2, wrk)
            ref = qo.cheby(psi0.copy(), Hh, 0.2, qo.ChebyWrk(psi0, 2.0 * bound, -bound, 0.2))
            assert np.linalg.norm(psi.numpy() - ref) < TOL
            rec.append(psi.numpy())
        outs.append(rec)
    ctx.tuning_set("value_dict", 1)
    assert all(np.array_equal(a, b) for a, b in zip(outs[0][1:], outs[1][1:]))
