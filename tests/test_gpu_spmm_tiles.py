"""GPU tests of the LDS-staged tiles of the batched Chebyshev term (csrc/kernels_spmm.hip: spmm_tile_kernel; plan: engine_core.hip
operator_spmm_tiles; BASELINE configs[4]): a lattice operator's interior rows are summed by workgroups that stage the 4 x 4 patch's
operands once -- the same entries in the same order as the row kernel, so the same bits; the reference would run `batch` independent
`cheby!` calls (src/cheby.jl:151-214), which is what the oracle does."""
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


def _steps(ctx, Op, states, batch, rw, strip=0, dts=(0.7, -0.7, 0.7)):
    N = states.shape[0]
    saved = {k: ctx.tuning_get(k) for k in ("spmm_rw", "spmm_strip")}
    try:
        ctx.tuning_set("spmm_rw", rw)
        ctx.tuning_set("spmm_strip", strip)
        info = Op.spmm_tiles(batch)
        wrk = L.ChebyWrk(ctx, N * batch, 20.0, -10.0, abs(dts[0]))
        panel = L.State(ctx, data=states.reshape(-1))
        for dt in dts:
            L.cheby_batched(panel, Op, dt, wrk, batch)
        out = panel.numpy().reshape(N, batch)
        panel.close()
        wrk.close()
    finally:
        for k, v in saved.items():
            ctx.tuning_set(k, v)
    return out, info


SHAPES = [  # (N, offsets, diagonal, batch): near reach NN, far reach K, patterns with gaps, panels that are no multiple of 64 states
    (8192, (1, 2, 3, 4, 256, 512, 768, 1024), False, 64),
    (8192, (1, 2, 3, 4, 256, 512, 768, 1024), True, 70),
    (6144 + 100, (1, 128), True, 64),                  # a five-point stencil; a ragged last strip step
    (8192, (1, 2, 128, 256), False, 40),
    (8192, (2, 4, 192, 576), True, 64),                # gaps: near {2, 4}, far {g, 3 g} with g = 192 (no multiple of 64)
    (16384, (3, 64, 128, 192, 256), False, 128),       # two slices of 64 states
    (8192, (128, 256), True, 64),                      # no near neighbours at all
]


@pytest.mark.parametrize("N,offsets,diag,batch", SHAPES)
def test_lds_tiles_bit_identical_to_the_row_kernel_and_match_the_oracle(ctx, N, offsets, diag, batch):
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets, rho=6.0)
    H = synth.to_scipy(rp, col, vals, N)
    if diag:
        H = sp.csr_matrix(H + sp.diags(np.random.default_rng(5).uniform(-1, 1, N)).astype(np.complex128))
        H.sort_indices()
    Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, H)])
    states = np.stack([synth.random_state(N, seed=4000 + s) for s in range(batch)], axis=1)
    rows, info0 = _steps(ctx, Op, states, batch, rw=0)
    assert info0["taken"] == 0
    for strip in (0, 32, -1):
        tiles, info = _steps(ctx, Op, states, batch, rw=-1, strip=strip)
        if strip < 0:
            assert info["taken"] == 0          # (spmm_strip -1: natural row order, no plan)
        else:
            g = int(np.gcd.reduce([d for d in offsets if d >= 64]))
            assert info["taken"] == 1 and info["g"] == g, info
            assert info["K"] == max(offsets) // g and info["NN"] == max([d for d in offsets if d <= 4], default=0)
            assert info["tiles"] * 16 + info["rest_rows"] == N and info["tiles"] * 16 >= N // 2
        assert np.array_equal(tiles, rows), strip
    for s in (0, batch - 1):
        owrk = qo.ChebyWrk(states[:, s].copy(), 20.0, -10.0, 0.7)
        ref = states[:, s].copy()
        for dt in (0.7, -0.7, 0.7):
            qo.cheby(ref, H, dt, owrk)
        assert np.linalg.norm(rows[:, s] - ref) < TOL, s


def test_lds_tiles_follow_the_coefficients_of_a_lazy_sum(ctx):
    """`evaluate!` between steps (src/generators.jl:757-766): the tile kernel reads the CSR-ordered mirror that is gathered again after
    qp_operator_set_coeffs; two controlled terms on a union pattern, a scale."""
    N, batch = 8192, 64
    mats, Hs = [], []
    for t, offs in enumerate(((1, 2, 256, 512), (1, 3, 256), (2, 512))):
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offs, rho=3.0, seed=70 + t)
        Hs.append(synth.to_scipy(rp, col, vals, N))
        mats.append(L.Matrix(ctx, N, N, rp, col, vals))
    Op = L.Operator(ctx, mats, 2)
    Op.set_scale(0.8)
    assert Op.spmm_tiles(batch)["taken"] == 1
    states = np.stack([synth.random_state(N, seed=4100 + s) for s in range(batch)], axis=1)
    wrk = L.ChebyWrk(ctx, N * batch, 20.0, -10.0, 0.4)
    panel = L.State(ctx, data=states.reshape(-1))
    refs = {s: states[:, s].copy() for s in (0, 31, 63)}
    for c1, c2 in ((1.0, 0.0), (0.3, -0.9), (-1.2, 0.5)):
        Op.set_coeffs([c1, c2])
        L.cheby_batched(panel, Op, 0.4, wrk, batch)
        H = sp.csr_matrix(0.8 * (Hs[0] + c1 * Hs[1] + c2 * Hs[2]))
        for s, r in refs.items():
            owrk = qo.ChebyWrk(r, 20.0, -10.0, 0.4)
            qo.cheby(r, H, 0.4, owrk)
    out = panel.numpy().reshape(N, batch)
    for s, r in refs.items():
        assert np.linalg.norm(out[:, s] - r) < TOL, s


def test_lds_tiles_not_taken_where_the_pattern_has_no_lattice(ctx):
    """A scattered pattern (no common inner dimension), a plain band (no far distances), distances between the near reach and 64,
    a far reach beyond four strip steps: the row kernel as before."""
    N = 1 << 14
    for offs in (synth.scattered_offsets(N), (1, 2, 3, 4, 5, 6, 7, 8), (1, 2, 20, 256), (1, 128, 256, 384, 512, 640)):
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offs)
        Op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
        assert Op.spmm_tiles(64)["taken"] == 0, offs
        Op.close()
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 128))
    Op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
    assert Op.spmm_tiles(64)["taken"] == 1 and Op.spmm_tiles(32)["taken"] == 0      # (panels of at most 32 states: the state-tiled kernel)


def test_lds_tiles_full_size_bit_identical_to_the_row_kernel(ctx):
    """BASELINE configs[4] at its full size (N = 2^18, 64 states): 15872 tiles + the 8192 rows of the first and last four strip steps
    (periodic wrap-around: other column order) by the row kernel -- the same bits as the row kernel alone; against the C port of the
    reference's serial path: tests/test_gpu_parity.py::test_batched_c5_full_size_properties (which now runs through the tiles too)."""
    N, b = 1 << 18, 64
    rp, col, vals = synth.hermitian_offsets_csr(N)
    Op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
    del rp, col, vals
    states = np.stack([synth.random_state(N, seed=500 + s) for s in range(b)], axis=1)
    tiles, info = _steps(ctx, Op, states, b, rw=-1, dts=(1.0,))
    assert info == {"taken": 1, "tiles": 15872, "rest_rows": 8192, "g": 1024, "K": 4, "NN": 4}
    rows, _ = _steps(ctx, Op, states, b, rw=0, dts=(1.0,))
    assert np.array_equal(tiles, rows)
    assert np.max(np.abs(np.linalg.norm(tiles, axis=0) - 1.0)) < 1e-11
