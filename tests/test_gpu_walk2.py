"""GPU parity tests of the TWO-TERM strip walk (csrc/kernels_walk2.hip: two fused Chebyshev terms of ``cheby!`` -- src/cheby.jl:186-209 --
per pass over the matrix values; knob walk_pair): bit-identical to one-term launches (the strip walk and the per-block kernel) for
every shape with a kernel instance, every cut of the walk, odd and even term counts, both time directions, lazy sums with
coefficients, real copies -- and within 1e-10 of the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-10
KNOBS = ("hrb_walk", "walk_pair", "walk_min_blocks", "walk_nt", "walk_waves")
DEFAULTS = {"hrb_walk": 1, "walk_pair": -1, "walk_min_blocks": 3072, "walk_nt": -1, "walk_waves": 0}


@pytest.fixture()
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _with_diagonal(rp, col, vals, N, seed=3):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    H = sp.csr_matrix((vals, col, rp), shape=(N, N)) + sp.diags(rng.uniform(-1.0, 1.0, N)).astype(np.complex128)
    H = H.tocsr()
    H.sort_indices()
    return H.indptr.astype(np.int64), H.indices.astype(np.int32), H.data.astype(np.complex128)


@pytest.mark.parametrize("N,offsets,diag,real,shape", [
    ((1 << 15) + 192, synth.BANDED_OFFSETS, False, False, (4, 4, 0)),      # the headline lattice (g = 1024), a partly filled last block
    (1 << 16, (1, 2, 3, 4, 64, 128, 192, 256), True, False, (4, 4, 1)),    # one row block per strip step, with a diagonal
    (1 << 15, (1, 2, 3, 4, 128, 256, 384, 512), False, True, (4, 4, 0)),   # real couplings: the real copy is streamed
    (1 << 15, (1, 3, 128, 256), False, False, (2, 2, 0)),
    (1 << 15, (2, 16, 192, 384), True, False, (2, 2, 1)),                  # near reach 16: chunks of 32 useful rows
    (1 << 14, (1, 64), True, False, (1, 1, 1)),                            # the five-point stencil of a periodic 64 x 256 grid
    (1 << 15, (5, 320), False, False, (1, 1, 0)),
    (1 << 16, (2, 5, 9, 320, 640, 960), False, False, (3, 3, 0)),
    (1 << 15, (1, 2, 3, 4, 256, 512), True, False, (4, 2, 1)),
    (1 << 15, (3, 128, 256, 384), False, False, (1, 3, 0)),
    (1 << 15, (1, 2, 7, 192), True, True, (3, 1, 1)),
    (1 << 16, (1, 6, 128, 256, 384, 512), False, False, (2, 4, 0)),
], ids=["headline", "g64+diag", "real", "near2far2", "near2far2+diag_d16", "5point", "near1far1_g320", "near3far3", "near4far2+diag",
        "near1far3", "near3far1_real+diag", "near2far4"])
def test_two_term_walk_bit_identical_to_one_term_launches(ctx, N, offsets, diag, real, shape):
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets)
    if real:
        vals = vals.real.astype(np.complex128)
    if diag:
        rp, col, vals = _with_diagonal(rp, col, vals, N)
    saved = {k: ctx.tuning_get(k) for k in KNOBS}
    try:
        ctx.tuning_set("walk_min_blocks", 16)
        Op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)], 0, L.FMT_HRB)
        wi = Op.walk_info()
        assert wi["valid"] == 1 and (wi["near"], wi["far"], wi["diag"]) == shape
        psi0 = synth.random_state(N)
        results = {}
        for dt in (1.0, 0.35):                 # 26 / 14 coefficients at Delta = 24: an odd and an even number of terms
            wrk = L.ChebyWrk(ctx, N, 24.0, -12.0, dt)

            def run(**knobs):
                for k, v in {**DEFAULTS, "walk_min_blocks": 16, **knobs}.items():
                    ctx.tuning_set(k, v)
                taken = Op.walk2_info()["valid"]
                psi = L.State(ctx, data=psi0)
                L.cheby(psi, Op, dt, wrk)
                L.cheby(psi, Op, -dt, wrk)
                L.cheby(psi, Op, dt, wrk)
                out = psi.numpy()
                psi.close()
                return out, taken

            base, t0 = run(hrb_walk=0)                                 # the per-block kernel, one term per launch
            one, t1 = run(walk_pair=0)                                 # the one-term strip walk
            assert t0 == 0 and t1 == 0 and np.array_equal(base, one)
            for knobs in (dict(walk_pair=1), dict(walk_pair=1, walk_waves=64), dict(walk_pair=1, walk_waves=4096),
                          dict(walk_pair=1, walk_waves=200, walk_nt=0), dict(walk_pair=1, walk_waves=1, walk_nt=1)):
                got, taken = run(**knobs)
                assert taken == 1, knobs
                assert np.array_equal(base, got), (dt, knobs, float(np.max(np.abs(base - got))))
            auto, taken = run()                                        # the automatic rule: not for an operator this small
            assert taken == 0 and np.array_equal(base, auto)
            results[dt] = base
            wrk.close()
        info = None
        ctx.tuning_set("walk_pair", 1)
        info = Op.walk2_info()
        assert info["valid"] == 1 and wi["first_block"] < info["first_block"] < info["end_block"] < wi["end_block"]
        assert info["edge_blocks"] == Op.layout_info()["blocks"] - (info["end_block"] - info["first_block"])
    finally:
        for k, v in saved.items():
            ctx.tuning_set(k, v)
    H = synth.to_scipy(rp, col, vals, N)
    ref = qo.cheby(psi0.copy(), H, 1.0, qo.ChebyWrk(psi0, 24.0, -12.0, 1.0))
    ref = qo.cheby(ref, H, -1.0, qo.ChebyWrk(psi0, 24.0, -12.0, 1.0))
    ref = qo.cheby(ref, H, 1.0, qo.ChebyWrk(psi0, 24.0, -12.0, 1.0))
    assert np.linalg.norm(results[1.0] - ref) < TOL


def test_two_term_walk_lazy_sum_and_check_normalization(ctx):
    """A lazy sum of two lattice terms with coefficients that change between steps (evaluate!, src/generators.jl:757-766) through the
    two-term walk against the oracle; with check_normalization the step falls back to one-term launches (the check needs every
    term's own launch) and still agrees bit for bit."""
    N = 1 << 15
    rp, col, v1 = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 128, 256, 384, 512))
    _, _, v2 = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 128, 256, 384, 512), seed=77)
    saved = {k: ctx.tuning_get(k) for k in KNOBS}
    try:
        ctx.tuning_set("walk_min_blocks", 16)
        ctx.tuning_set("walk_pair", 1)
        Op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, v1), L.Matrix(ctx, N, N, rp, col, v2)], 1, L.FMT_HRB)
        psi0 = synth.random_state(N)
        wrk = L.ChebyWrk(ctx, N, 50.0, -25.0, 0.4)
        psi = L.State(ctx, data=psi0)
        ref = psi0.copy()
        owrk = qo.ChebyWrk(psi0, 50.0, -25.0, 0.4)
        for cval in (0.7, -0.3, 1.2):
            Op.set_coeffs([cval])
            assert Op.walk2_info()["valid"] == 1
            L.cheby(psi, Op, 0.4, wrk)
            H = synth.to_scipy(rp, col, v1 + cval * v2, N)
            qo.cheby(ref, H, 0.4, owrk)
            assert np.linalg.norm(psi.numpy() - ref) < TOL, cval
        a = psi.numpy()
        psi2 = L.State(ctx, data=psi0)
        for cval in (0.7, -0.3, 1.2):
            Op.set_coeffs([cval])
            L.cheby(psi2, Op, 0.4, wrk, check_normalization=True)
        assert np.array_equal(a, psi2.numpy())
    finally:
        for k, v in saved.items():
            ctx.tuning_set(k, v)


def test_two_term_walk_inside_the_fused_propagate_loop(ctx):
    """`propagate` with the step loop inside the library (qp_propagate: N1 of SURVEY 8f) and a time-dependent control: every step's
    evaluate! + cheby! with the terms in pairs gives the bits of the one-term walk, states stored after every step included."""
    import qprop_amd.propagator as P
    N = 1 << 15
    rp, col, v1 = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 128, 256, 384, 512))
    _, _, v2 = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 128, 256, 384, 512), seed=78)
    H0, H1 = synth.to_scipy(rp, col, v1, N), synth.to_scipy(rp, col, v2, N)
    tlist = np.linspace(0.0, 0.6, 7)
    psi0 = synth.random_state(N)
    outs = {}
    saved = {k: ctx.tuning_get(k) for k in ("walk_pair", "walk_min_blocks")}
    try:
        ctx.tuning_set("walk_min_blocks", 16)
        for pair in (0, 1):
            ctx.tuning_set("walk_pair", pair)
            psi_T, states = P.propagate(psi0, P.hamiltonian(H0, (H1, lambda t: 0.5 * np.cos(4.0 * t))), tlist, method="cheby", storage=True,
                                        ctx=ctx, specrange_method="manual", E_min=-16.0, E_max=16.0)
            outs[pair] = (np.asarray(psi_T), np.asarray(states))
    finally:
        for k, v in saved.items():
            ctx.tuning_set(k, v)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert abs(np.linalg.norm(outs[1][0]) - 1.0) < 1e-11


def test_two_term_walk_full_size_against_the_c_port():
    """N = 2^22 (beyond the Infinity Cache: the AUTOMATIC rule takes the two-term walk): two steps forward and one back against the one-term
    walk (bit for bit) and, on a window of 2^18 rows, against the C port of the reference (a row depends on the rows within
    31 x 4096 of it after one step: the middle of the window is exact)."""
    from oracle import ref_c
    N = 1 << 22
    c = L.Context(0)
    try:
        rp, col, vals = synth.hermitian_offsets_csr(N)
        Op = L.Operator(c, [L.Matrix(c, N, N, rp, col, vals)])
        del rp, col, vals
        assert Op.walk_info()["valid"] == 1 and Op.walk2_info()["valid"] == 1
        psi0 = synth.random_state(N)
        wrk = L.ChebyWrk(c, N, 20.0, -10.0, 1.0)
        psi = L.State(c, data=psi0)
        L.cheby(psi, Op, 1.0, wrk)
        one = psi.numpy()
        L.cheby(psi, Op, 1.0, wrk)
        L.cheby(psi, Op, -1.0, wrk)
        pair = psi.numpy()
        c.tuning_set("walk_pair", 0)
        assert Op.walk2_info()["valid"] == 0
        psi.upload(psi0)
        L.cheby(psi, Op, 1.0, wrk)
        assert np.array_equal(one, psi.numpy())
        L.cheby(psi, Op, 1.0, wrk)
        L.cheby(psi, Op, -1.0, wrk)
        assert np.array_equal(pair, psi.numpy())
        assert abs(np.linalg.norm(pair) - 1.0) < 1e-11 and np.linalg.norm(pair - one) < 1e-10      # forward + backward = identity
        # window [lo, lo + W) of the operator against the C port
        W, half, lo = 1 << 18, 2048, (1 << 21) - (1 << 17) + 12345 * 64
        coeffs = L.cheby_coeffs(20.0, 1.0)
        assert (len(coeffs) - 1) * 4096 + half < W // 2
        wrp, wcol, wval = synth.hermitian_offsets_csr(N, row_begin=lo, row_end=lo + W)
        wloc = wcol.astype(np.int64) - lo
        keep = (wloc >= 0) & (wloc < W)
        lens = np.add.reduceat(keep.astype(np.int64), np.arange(0, len(keep), 16))
        wrp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        wpsi = psi0[lo:lo + W].copy()
        ref_c.cheby_csc(wrp2, wloc[keep], np.conj(wval[keep]), wpsi, coeffs, 20.0, -10.0, 1.0)     # Hermitian window block: CSC = conj CSR
        mid = slice(W // 2 - half, W // 2 + half)
        assert float(np.max(np.abs(one[lo:lo + W][mid] - wpsi[mid]))) < 1e-12
    finally:
        c.close()
