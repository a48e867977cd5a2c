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
