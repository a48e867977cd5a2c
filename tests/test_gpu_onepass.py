"""GPU parity tests of the Arnoldi sweep that reads the Krylov basis ONCE per column (knob arnoldi_onepass = 2,
csrc/kernels_onepass.hip) inside newton! (src/newton.jl:246-385 with src/arnoldi.jl:60-100 under it): against the oracle after
every step to 1e-10, the oracle's restart counts, Krylov breakdown, the eigenstate shortcut, real-valued operators, custom
functions, both time directions -- and that the sweep really is the one-pass one (launches per column)."""
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
import qprop_amd.synth as synth  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture()
def ctx():
    c = L.Context(0)
    c.tuning_set("arnoldi_onepass", 2)
    yield c
    c.close()


def _rand_state(N, rng):
    psi = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    return psi / np.linalg.norm(psi)


def _steps(ctx, A, psi0, dts, m_max, func=None, expect_onepass=True, **kw):
    """newton! step by step on the device and in the oracle; returns the device workspace of the last step."""
    N = len(psi0)
    A = sp.csr_matrix(A)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, A)], 0, L.FMT_RBCSR)
    wrk = L.NewtonWrk(ctx, N, m_max=m_max)
    psi = L.State(ctx, data=psi0)
    owrk = qo.NewtonWrk(psi0, m_max=m_max)
    ref = psi0.copy()
    pyf = {None: None, "exp": np.exp}.get(func, func)
    for k, dt in enumerate(dts):
        ctx.reset_stats()
        L.newton(psi, op, dt, wrk, func=func, **kw)
        qo.newton(ref, A, dt, owrk, func=pyf, **kw)
        assert np.linalg.norm(psi.numpy() - ref) < TOL * max(1.0, np.linalg.norm(ref)), (k, dt)
        assert wrk.restarts == owrk.restarts, (k, wrk.restarts, owrk.restarts)
        if expect_onepass and wrk.stats["n_matvec"] >= 4:
            # column kernel + solve per column, one more pair per sweep for the last vector, the combine and a copy or two per
            # restart: far below the 2 launches per column + projection of the two-pass sweep only if the path was taken --
            # what tells them apart is that NO launch of the two-pass kernels happened: the count is exactly determined
            sweeps = wrk.restarts + 1
            assert ctx.stats()["n_kernel_launches"] <= 2 * (wrk.stats["n_matvec"] + sweeps) + 3 * sweeps + 2, ctx.stats()
    return wrk, owrk


@pytest.mark.parametrize("n,m,dt", [(40, 6, 0.5), (96, 12, 0.5), (96, 20, 0.5), (128, 20, 1.0), (200, 17, 0.3)])
def test_onepass_newton_liouvillian(ctx, n, m, dt):
    """BASELINE config C3's operator at oracle sizes (N = n^2 up to 40000: two rounds of row blocks per wavefront, a partly filled
    last block), m_max up to the largest the kernels take, forward and backward."""
    Lm = synth.liouvillian_tridiag(n)
    rho0 = synth.random_state(Lm.shape[0])
    wrk, owrk = _steps(ctx, Lm, rho0, (dt, dt, -dt), m)
    assert wrk.restarts >= 0


def test_onepass_newton_hermitian_real_and_complex(ctx):
    """A banded Hermitian operator (complex values) and its real part (the real copy of the values is streamed), N = 2^15."""
    N = 1 << 15
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    H = synth.to_scipy(rp, col, vals, N)
    psi0 = synth.random_state(N)
    _steps(ctx, H, psi0, (0.3, 0.3, -0.3), 10)
    Hr = sp.csr_matrix((vals.real.astype(np.complex128), col, rp), shape=(N, N))
    _steps(ctx, Hr, psi0, (0.3, -0.3), 12)


def test_onepass_newton_custom_func_and_dense_exp(ctx):
    """test/test_newton.jl:130-177 in spirit: func = exp on a sparse non-Hermitian generator large enough for the multi-launch
    sweep, against dense exp."""
    rng = np.random.default_rng(44)
    N = 3000
    A = sp.random(N, N, density=0.003, random_state=7, format="csr", dtype=np.float64)
    A = (A + 1j * sp.random(N, N, density=0.003, random_state=8, format="csr", dtype=np.float64)).tocsr()
    A = A * (3.0 / abs(A).sum(axis=1).max()) - 0.2 * sp.identity(N)
    psi0 = _rand_state(N, rng)
    wrk, _ = _steps(ctx, A, psi0, (0.5,), 20, func="exp", max_restarts=40)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, sp.csr_matrix(A))], 0, L.FMT_RBCSR)
    psi = L.State(ctx, data=psi0)
    L.newton(psi, op, 0.5, L.NewtonWrk(ctx, N, m_max=20), func="exp", max_restarts=40)
    assert np.linalg.norm(psi.numpy() - sla.expm(A.toarray() * 0.5) @ psi0) < TOL


@pytest.mark.parametrize("pipeline", [1, 0])
def test_onepass_breakdown(ctx, pipeline):
    """Krylov space exhausted (src/arnoldi.jl:91-95): four distinct eigenvalues.  The column kernels after the breakdown run on
    (and are discarded), m shrinks for the later restarts, the vector of the breakdown enters with the reference's weight."""
    N = 20000
    rng = np.random.default_rng(5)
    lam = np.array([-3.0, 0.5, 2.0, 7.5])
    d = lam[rng.integers(0, 4, N)]
    A = sp.diags([d], [0], format="csr", dtype=complex)
    psi0 = _rand_state(N, rng)
    ctx.tuning_set("newton_pipeline", pipeline)
    wrk, owrk = _steps(ctx, A, psi0, (0.7, 0.7), 10, norm_min=1e-9)
    assert wrk.stats["n_matvec"] == 14      # a sweep of ten columns that breaks down at the fourth, then one of four


def test_onepass_eigenstate_shortcut(ctx):
    """src/newton.jl:289-295: an eigenvector as the start vector -- m = 1 in the first sweep, psi is scaled by f(lambda)."""
    N = 5000
    rng = np.random.default_rng(9)
    d = rng.uniform(-2, 2, N)
    A = sp.diags([d], [0], format="csr", dtype=complex)
    e = np.zeros(N, dtype=complex)
    e[1234] = 1.0
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, A)], 0, L.FMT_RBCSR)
    wrk = L.NewtonWrk(ctx, N, m_max=8)
    psi = L.State(ctx, data=e)
    L.newton(psi, op, 0.9, wrk)
    assert wrk.restarts == 0 and np.linalg.norm(psi.numpy() - np.exp(-0.9j * d[1234]) * e) < 1e-13


def test_onepass_is_deterministic_and_matches_two_pass_to_rounding(ctx):
    """Run-to-run identical bits (check_propagator's reinit test, src/interfaces/propagator.jl, asks for 1e-14), and the
    one-pass result within rounding of the low-synchronisation sweep's."""
    Lm = synth.liouvillian_tridiag(128)
    N = Lm.shape[0]
    rho0 = synth.random_state(N)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lm)])
    outs = []
    for mode in (2, 2, 0):
        ctx.tuning_set("arnoldi_onepass", mode)
        wrk = L.NewtonWrk(ctx, N, m_max=20)
        rho = L.State(ctx, data=rho0)
        for _ in range(3):
            L.newton(rho, op, 0.5, wrk)
        outs.append(rho.numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.linalg.norm(outs[0] - outs[2]) < 1e-12


def test_onepass_sweep_redone_in_two_pass_form_on_norm_drift(ctx):
    """the one-pass sweep carries H a_t forward by linearity and cannot see its own error growth; the host checks the
    stored basis vectors' measured norms after every such sweep and does the sweep AGAIN in the two-pass form when one drifted
    from 1 by more than 1e-4.  Knob value 3 forces that path on every sweep: same parity, same restart counts, and the step's
    statistics say how many sweeps were redone (none at knob value 2 on a benign operator)."""
    Lm = synth.liouvillian_tridiag(96)
    rho0 = synth.random_state(Lm.shape[0])
    wrk, owrk = _steps(ctx, Lm, rho0, (0.5, -0.5), 12)
    assert wrk.stats["sweeps_onepass"] == wrk.restarts + 1 and wrk.stats["sweeps_onepass_redone"] == 0
    ctx.tuning_set("arnoldi_onepass", 3)
    wrk, owrk = _steps(ctx, Lm, rho0, (0.5, 0.5, -0.5), 12, expect_onepass=False)
    assert wrk.stats["sweeps_onepass"] == wrk.restarts + 1 == wrk.stats["sweeps_onepass_redone"]
    ctx.tuning_set("arnoldi_onepass", 0)
    wrk, owrk = _steps(ctx, Lm, rho0, (0.5,), 12, expect_onepass=False)
    assert wrk.stats["sweeps_onepass"] == 0 and wrk.stats["sweeps_onepass_redone"] == 0


def test_onepass_sweep_is_taken_automatically_beyond_the_infinity_cache():
    """The AUTOMATIC branch (knob at its default, 1): a Liouvillian with basis + matrix above 224 MB (n = 800: N = 640 000, m_max = 20:
    16 N 23 + matrix = 0.29 GB) takes the one-pass sweep by itself -- statistics say so -- and agrees with the oracle after the step
    (1e-10), restart counts equal; one size below the threshold (n = 512, config C3) does not take it."""
    c = L.Context(0)
    try:
        assert c.tuning_get("arnoldi_onepass") == 1
        for n, expect in ((800, True), (512, False)):
            Lm = synth.liouvillian_tridiag(n)
            N = Lm.shape[0]
            rho0 = synth.random_state(N)
            op = L.Operator(c, [L.Matrix.from_scipy(c, Lm)])
            wrk = L.NewtonWrk(c, N, m_max=20)
            psi = L.State(c, data=rho0)
            L.newton(psi, op, 0.5, wrk)
            assert (wrk.stats["sweeps_onepass"] == wrk.restarts + 1) if expect else (wrk.stats["sweeps_onepass"] == 0), (n, wrk.stats)
            assert wrk.stats["sweeps_onepass_redone"] == 0
            if expect:
                owrk = qo.NewtonWrk(rho0, m_max=20)
                ref = qo.newton(rho0.copy(), Lm, 0.5, owrk)
                assert np.linalg.norm(psi.numpy() - ref) < TOL and wrk.restarts == owrk.restarts
            for h in (psi, wrk, op):
                h.close()
    finally:
        c.close()
