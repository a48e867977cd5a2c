"""GPU tests of the host mirror of the reference's propagator interface
(init_prop / prop_step! / reinit_prop! / set_state! / set_t! / propagate), restating the
reference's own interface tests: test/test_propagate.jl:74-163, test/test_prop_interfaces.jl
and the `check_propagator` contract (src/interfaces/propagator.jl:55-338)."""
import os
import sys
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.propagator as P  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _optomech():
    """test/optomech.jl:1-44 (deterministic)."""
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
    psi0[2] = 1.0
    return H.tocsr(), psi0


def test_tls_rabi_verbatim(ctx):
    """test/test_propagate.jl:10-71, :74-150 with the reference's own literals through the device path:
    `(H,)` tuple generator, H = [[0, 0.5], [0.5, 0]], tlist = range(0, 1.5 pi, length = 101), expected
    [-1/sqrt 2, -i/sqrt 2], forward + backward 1e-12, storage forward == storage backward 1e-12;
    in place (mutable state, :10-71) and not in place ("Immutable TLS", :74-150)."""
    psi0 = np.array([1, 0], dtype=complex)
    H = np.array([[0, 0.5], [0.5, 0]], dtype=complex)
    tlist = np.linspace(0, 1.5 * np.pi, 101)
    generator = (H,)
    expected = np.array([-1 / np.sqrt(2), -1j / np.sqrt(2)])
    for inplace in (True, False):
        out, storage = P.propagate(psi0, generator, tlist, method="cheby", inplace=inplace, storage=True, ctx=ctx)
        assert np.linalg.norm(out - expected) < 1e-12
        pop0 = np.abs(storage[0, :]) ** 2
        assert abs(pop0[-1] - 0.5) < 1e-8 and abs(pop0[0] - 1.0) < 1e-15
        back, storage_bw = P.propagate(out, generator, tlist, method="cheby", backward=True, inplace=inplace,
                                       storage=True, ctx=ctx)
        assert np.linalg.norm(back - psi0) < 1e-12
        assert abs(abs(storage_bw[0, 0]) ** 2 - 1.0) < 1e-8
        assert np.linalg.norm(storage - storage_bw) < 1e-12
    # the oracle and the device agree on every stored state
    _, ostore = qo.propagate(psi0, generator, tlist, "cheby", storage=True)
    assert np.linalg.norm(storage - ostore) < 1e-12


def test_tls_rabi(ctx):
    """test/test_propagate.jl:74-150: Cheby forward/backward vs analytic, 1e-12; both
    in-place and not-in-place (a new state object per step)."""
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    sz = np.array([[1, 0], [0, -1]], dtype=complex)
    T = np.pi / 2
    tlist = np.linspace(0, T, 101)
    gen = P.hamiltonian(0.0 * sz, (0.5 * sx, lambda t: 1.0))
    psi0 = np.array([1, 0], dtype=complex)
    expected = np.array([1 / np.sqrt(2), -1j / np.sqrt(2)])
    for inplace in (True, False):
        out = P.propagate(psi0, gen, tlist, method="cheby", inplace=inplace, ctx=ctx)
        assert np.linalg.norm(out - expected) < 1e-12
        back = P.propagate(out, gen, tlist, method="cheby", backward=True, inplace=inplace, ctx=ctx)
        assert np.linalg.norm(back - psi0) < 1e-12
    # Newton on a two-level system is rejected, as in the reference (src/newton.jl:41-46)
    with pytest.raises(L.QPArgumentError, match="state dimension > 2"):
        P.propagate(psi0, gen, tlist, method="newton", ctx=ctx)


def test_optomech_newton_vs_cheby(ctx):
    """test/test_propagate.jl:153-163: generator `(H,)`, tlist 0:0.2:50, Newton norm,
    Newton vs Cheby < 1e-10 (specrange :auto -> :arnoldi for N=55), and vs the oracle."""
    H, psi0 = _optomech()
    tlist = np.arange(0, 50 + 1e-9, 0.2)
    psi1 = P.propagate(psi0, (H,), tlist, method="newton", ctx=ctx)
    assert (np.linalg.norm(psi1) - 1.0) < 1e-12
    psi2 = P.propagate(psi0, (H,), tlist, method="cheby", ctx=ctx, rng=np.random.default_rng(1))
    assert np.linalg.norm(psi1 - psi2) < 1e-10
    ref = qo.propagate(psi0, H, tlist, "newton")
    assert np.linalg.norm(psi1 - ref) < 1e-10


def test_time_dependent_generator_matches_oracle(ctx):
    """Drift + two controls, values rewritten every step (device `evaluate!`), storage."""
    rng = np.random.default_rng(3)
    N = 96
    H0 = synth.dense_hermitian(N, rho=3.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    H2 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    tlist = np.linspace(0, 2.0, 41)
    e1 = lambda t: 0.8 * np.sin(3 * t)        # noqa: E731
    e2 = np.cos(np.linspace(0, 1, 40)) * 0.5    # defined on the intervals
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    for method, kw in (("cheby", dict(E_min=-8.0, E_max=8.0)), ("newton", dict(m_max=8))):
        out, store = P.propagate(psi0, P.hamiltonian(H0, (H1, e1), (H2, e2)), tlist, method=method,
                                 storage=True, ctx=ctx, **kw)
        ref, rstore = qo.propagate(psi0, qo.Generator([H0, H1, H2], [e1, e2]), tlist, method, storage=True, **kw)
        assert np.linalg.norm(out - ref) < 1e-10
        assert np.max(np.linalg.norm(store - rstore, axis=0)) < 1e-10


def _check_propagator(p, psi0, atol=1e-14):
    """The acceptance contract of src/interfaces/propagator.py:55-338, restated."""
    tlist = p.tlist
    assert p.t == (tlist[-1] if p.backward else tlist[0])
    s0 = p.state
    s1 = P.prop_step(p)
    if p.inplace:
        assert s1 is s0                       # identical object when in-place
    else:
        assert s1 is not s0                   # a new object when not
    assert s1 is p.state
    assert abs(p.t - (tlist[-2] if p.backward else tlist[1])) < atol
    psi1 = s1.numpy()
    assert abs(np.linalg.norm(psi1) - 1) < 1e-10
    with pytest.raises(AttributeError):
        p.generator
    # run to the end: prop_step! returns nothing and leaves the propagator unchanged
    while P.prop_step(p) is not None:
        pass
    assert p.t == (tlist[0] if p.backward else tlist[-1])
    n_end, t_end = p.n, p.t
    assert P.prop_step(p) is None and p.n == n_end and p.t == t_end
    # set_t! / set_state!
    P.set_t(p, tlist[3])
    assert p.t == tlist[3]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        P.set_t(p, 0.5 * (tlist[3] + tlist[4]) + 1e-3)
        assert any("Snapping" in str(x.message) for x in w)
    ret = P.set_state(p, psi0)
    assert ret is p.state and np.array_equal(p.state.numpy(), psi0)
    for par in p.parameters:
        assert len(par) == len(tlist) - 1
    # reinit_prop!: re-running the first step reproduces Psi_1 to 1e-14
    P.reinit_prop(p, psi0)
    assert p.t == (tlist[-1] if p.backward else tlist[0])
    again = P.prop_step(p).numpy()
    assert np.linalg.norm(again - psi1) < 1e-14


@pytest.mark.parametrize("method,inplace,backward", [
    ("cheby", True, False), ("cheby", False, False), ("cheby", True, True),
    ("newton", True, False), ("newton", True, True)])
def test_check_propagator_contract(ctx, method, inplace, backward):
    """test/test_prop_interfaces.jl:12-103."""
    rng = np.random.default_rng(5)
    N = 40
    H0 = synth.dense_hermitian(N, rho=2.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    tlist = np.linspace(0, 1.0, 11)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    p = P.init_prop(psi0, P.hamiltonian(H0, (H1, lambda t: np.sin(t))), tlist, method, inplace=inplace,
                    backward=backward, ctx=ctx, rng=np.random.default_rng(0))
    _check_propagator(p, psi0)


def test_interface_errors(ctx):
    """test/test_prop_interfaces.jl:79-89, 309-412."""
    H = synth.dense_hermitian(40, rho=2.0, rng=np.random.default_rng(6))
    psi0 = np.zeros(40, dtype=complex)
    psi0[0] = 1
    tlist = np.linspace(0, 1, 11)
    with pytest.raises(RuntimeError, match="The Newton propagator is only implemented in-place"):
        P.init_prop(psi0, H, tlist, "newton", inplace=False, ctx=ctx)
    with pytest.raises(ValueError, match="Unknown propagation `method`"):
        P.init_prop(psi0, H, tlist, "foo", ctx=ctx)
    bad = np.array([0, 0.1, 0.2, 0.35, 0.5])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="uniform time grid"):
            P.init_prop(psi0, H, bad, "cheby", ctx=ctx)
    p = P.init_prop(psi0, H, bad, "newton", ctx=ctx)      # Newton takes non-uniform grids
    while P.prop_step(p) is not None:
        pass
    assert abs(p.state.norm() - 1) < 1e-12


def test_cheby_envelope_and_reinit(ctx):
    """test/test_specrad.jl:147-223 + reinit coefficient refresh (cheby_propagator.jl:243-299)."""
    rng = np.random.default_rng(7)
    N = 64
    H0 = synth.dense_hermitian(N, rho=3.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    tlist = np.linspace(0, 1, 11)
    p = P.init_prop(psi0, H0, tlist, "cheby", ctx=ctx, E_min=-10, E_max=10)
    assert abs(p.wrk.E_min + 10.1) < 1e-12 and abs(p.wrk.Delta - 20.2) < 1e-12
    p = P.init_prop(psi0, H0, tlist, "cheby", ctx=ctx, E_min=-10, E_max=10, specrange_buffer=0.1)
    assert abs(p.wrk.E_min + 11.0) < 1e-12 and abs(p.wrk.Delta - 22.0) < 1e-12
    # arnoldi envelope brackets the true spectrum of both extremal operators
    amp = np.linspace(-1, 1, 10)
    p = P.init_prop(psi0, P.hamiltonian(H0, (H1, amp)), tlist, "cheby", ctx=ctx, rng=np.random.default_rng(2))
    for cval in (-1, 1):
        ev = np.linalg.eigvalsh(H0 + cval * H1)
        assert p.wrk.E_min <= ev[0] and ev[-1] <= p.wrk.E_min + p.wrk.Delta
    n_before, D_before = p.wrk.n_coeffs, p.wrk.Delta
    P.reinit_prop(p, psi0)                       # ranges unchanged: same workspace
    assert p.wrk.n_coeffs == n_before
    p.parameters[0] = p.parameters[0] * 5.0      # controls grew: coefficients are refreshed
    P.reinit_prop(p, psi0)
    assert p.wrk.Delta > D_before
    while P.prop_step(p) is not None:
        pass
    ref = qo.propagate(psi0, qo.Generator([H0, H1], [amp * 5.0]), tlist, "cheby", E_min=p.wrk.E_min,
                       E_max=p.wrk.E_min + p.wrk.Delta, specrange_buffer=0.0)
    assert np.linalg.norm(p.state.numpy() - ref) < 1e-10


@pytest.mark.parametrize("method", ["cheby", "newton"])
@pytest.mark.parametrize("backward", [False, True])
def test_fused_propagate_equals_stepwise(ctx, method, backward):
    """qp_propagate (the step loop inside the library, src/propagate.jl:283-344) gives
    bit-identical states to calling prop_step! from the host; matrix observables are
    dot(psi, O, psi) (src/storage.jl:121-123); backward storage is filled from the end."""
    rng = np.random.default_rng(11)
    N = 80
    H0 = synth.dense_hermitian(N, rho=3.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    tlist = np.linspace(0, 1.5, 31)
    if method == "newton":                      # Newton also takes a non-uniform grid
        tlist = np.cumsum(np.concatenate([[0.0], 0.03 + 0.04 * rng.random(30)]))
    eps = lambda t: 0.7 * np.cos(2 * t)        # noqa: E731
    gen = P.hamiltonian(H0, (H1, eps))
    kw = dict(E_min=-6.0, E_max=6.0) if method == "cheby" else dict(m_max=7)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    L.tuning_set("small_nnz", 0)                # the general loop on both sides: one launch per term / per
    try:                                        # Arnoldi step (the persistent kernels round differently)
        out_f, st_f = P.propagate(psi0, gen, tlist, method=method, backward=backward, storage=True, ctx=ctx,
                                  fused=True, **kw)
        out_s, st_s = P.propagate(psi0, gen, tlist, method=method, backward=backward, storage=True, ctx=ctx,
                                  fused=False, **kw)
    finally:
        L.tuning_set("small_nnz", 8192)
    assert np.array_equal(out_f, out_s) and np.array_equal(st_f, st_s)
    # default: small Cheby systems run the whole grid in one persistent launch
    out_p, st_p = P.propagate(psi0, gen, tlist, method=method, backward=backward, storage=True, ctx=ctx, **kw)
    assert np.max(np.linalg.norm(st_p - st_s, axis=0)) < 1e-13 and np.linalg.norm(out_p - out_s) < 1e-13
    ref, rstore = qo.propagate(psi0, qo.Generator([H0, H1], [eps]), tlist, method, backward=backward,
                               storage=True, **kw)
    assert np.max(np.linalg.norm(st_f - rstore, axis=0)) < 1e-10
    # observables: one dense Hermitian, one sparse non-Hermitian matrix
    O1 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    O2 = sp.random(N, N, density=0.1, random_state=5, format="csr") * (1 + 0.5j)
    out_o, ev = P.propagate(psi0, gen, tlist, method=method, backward=backward, storage=True,
                            observables=[O1, O2], ctx=ctx, **kw)
    assert np.array_equal(out_o, out_p) and ev.shape == (2, len(tlist))
    want = np.array([[np.vdot(st_p[:, i], O @ st_p[:, i]) for i in range(len(tlist))] for O in (O1, O2)])
    assert np.max(np.abs(ev - want)) < 1e-12
    # without storage only the final state comes back
    assert np.array_equal(P.propagate(psi0, gen, tlist, method=method, backward=backward, ctx=ctx, **kw), out_p)
    with pytest.raises(ValueError):
        P.propagate(psi0, gen, tlist, method=method, ctx=ctx, fused=True, callback=lambda *a: None, **kw)


@pytest.mark.parametrize("N,dense,ncontrols", [(2, True, 1), (3, True, 0), (55, False, 0), (64, True, 2),
                                               (100, True, 1), (128, True, 2), (130, True, 1), (700, False, 2), (2048, False, 1), (3000, False, 1)])
def test_persistent_small_cheby(ctx, N, dense, ncontrols):
    """The single-launch time grid for small systems (register-resident rows, LDS vectors,
    streaming rows, vectors in global memory -- one case each) against the oracle, the
    general loop, with observables, backward, and the normalization check."""
    rng = np.random.default_rng(N)
    if dense:
        mats = [synth.dense_hermitian(N, rho=2.0 if i == 0 else 0.5, rng=rng) for i in range(ncontrols + 1)]
    else:
        mats = [synth.sparse_random(N, min(1.0, 12.0 / N), rho=2.0 if i == 0 else 0.5, hermitian=True, rng=rng)
                for i in range(ncontrols + 1)]
    nt = 25
    tlist = np.linspace(0, 1.0, nt)
    ctrls = [(lambda t, k=k: 0.6 * np.sin((k + 2) * t + 0.3)) for k in range(ncontrols)]
    gen = P.hamiltonian(mats[0], *[(m, c) for m, c in zip(mats[1:], ctrls)]) if ncontrols else (mats[0],)
    ogen = qo.Generator(mats, ctrls) if ncontrols else mats[0]
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    kw = dict(E_min=-4.0, E_max=4.0)
    for backward in (False, True):
        out, st = P.propagate(psi0, gen, tlist, method="cheby", storage=True, backward=backward, ctx=ctx, **kw)
        ref, rst = qo.propagate(psi0, ogen, tlist, "cheby", storage=True, backward=backward, **kw)
        assert np.max(np.linalg.norm(st - rst, axis=0)) < 1e-10 and np.linalg.norm(out - ref) < 1e-10
        L.tuning_set("small_nnz", 0)
        try:
            out_g = P.propagate(psi0, gen, tlist, method="cheby", backward=backward, ctx=ctx, **kw)
        finally:
            L.tuning_set("small_nnz", 8192)
        assert np.linalg.norm(out - out_g) < 1e-13
    O = mats[0]
    _, ev = P.propagate(psi0, gen, tlist, method="cheby", storage=True, observables=[O], ctx=ctx, **kw)
    O_ = O.toarray() if sp.issparse(O) else O
    _, st = P.propagate(psi0, gen, tlist, method="cheby", storage=True, ctx=ctx, **kw)
    want = np.array([np.vdot(st[:, i], O_ @ st[:, i]) for i in range(nt)])
    assert np.max(np.abs(ev[0] - want)) < 1e-12
    # spectral range too narrow: "Incorrect normalization" surfaces from inside the launch
    with pytest.raises(L.QPError, match="Incorrect normalization"):
        P.propagate(psi0, gen, tlist, method="cheby", ctx=ctx, E_min=3.0, E_max=3.2, check_normalization=True)
    # and the operator is left holding the last interval's coefficients
    P.propagate(psi0, gen, tlist, method="cheby", ctx=ctx, check_normalization=True, **kw)


def test_cheby_graph_replay_is_bit_identical(ctx):
    """Knob `cheby_graph`: a repeated cheby! step replayed as a hipGraph enqueues exactly the
    same launches; the counters of qp_stats advance as if they had been launched one by one."""
    rng = np.random.default_rng(21)
    N = 1500
    H = synth.sparse_random(N, 10.0 / N, rho=2.0, hermitian=True, rng=rng)
    tlist = np.linspace(0, 1.0, 21)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    kw = dict(E_min=-4.0, E_max=4.0, fused=False)
    ctx.reset_stats()
    ref = P.propagate(psi0, (H,), tlist, method="cheby", ctx=ctx, **kw)
    plain = ctx.stats()
    L.tuning_set("cheby_graph", 1024)
    try:
        ctx.reset_stats()
        out = P.propagate(psi0, (H,), tlist, method="cheby", ctx=ctx, **kw)
        st = ctx.stats()
    finally:
        L.tuning_set("cheby_graph", 0)
    assert np.array_equal(out, ref)
    assert st["n_graph_launches"] == len(tlist) - 2        # the first call only arms the key
    assert st["n_matvec"] == plain["n_matvec"] and st["spmv_bytes"] == plain["spmv_bytes"]


def test_interfaces_device_types(ctx):
    """QuantumPropagators.Interfaces.check_state / check_operator / check_propagator for the
    device-resident types (src/interfaces/*.jl; test/test_prop_interfaces.jl and the
    `check_*` calls of test/test_propagate.jl), including that the checks do catch a type
    that breaks the contract."""
    import qprop_amd.interfaces as I
    rng = np.random.default_rng(31)
    N = 300
    psi = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi /= np.linalg.norm(psi)
    state = L.State(ctx, data=psi)
    assert I.check_state(state, normalized=True)
    assert not I.check_state(2.0 * state, normalized=True, quiet=True)
    for A in (synth.dense_hermitian(N, rho=3.0, rng=rng), synth.sparse_random(N, 0.05, rho=3.0, rng=rng)):
        Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, sp.csr_matrix(A))])
        assert I.check_operator(Op, state=state, atol=1e-12)
        A_ = A.toarray() if sp.issparse(A) else A
        assert np.linalg.norm((Op * state).numpy() - A_ @ psi) < 1e-12
        assert abs(Op.dot(state, state) - np.vdot(psi, A_ @ psi)) < 1e-12

    class Broken(L.State):                        # axpy! that forgets the factor
        def axpy(self, alpha, x):
            return L.State.axpy(self, 1.0, x)

        def similar(self):
            return Broken(self.ctx, n=self.n)

        def copy(self):
            return Broken(self.ctx, n=self.n).copy_from(self)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert not I.check_state(Broken(ctx, data=psi))
        assert any("axpy!" in str(x.message) for x in w)
    H0 = synth.dense_hermitian(N, rho=3.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    tlist = np.linspace(0, 1.0, 11)
    for method, kw in (("cheby", dict(E_min=-6.0, E_max=6.0)), ("newton", dict(m_max=6))):
        for inplace, backward in ((True, False), (True, True)) + (((False, False),) if method == "cheby" else ()):
            p = P.init_prop(psi, P.hamiltonian(H0, (H1, lambda t: 0.3 * t)), tlist, method, ctx=ctx,
                            inplace=inplace, backward=backward, **kw)
            assert I.check_propagator(p, atol=1e-13)


@pytest.mark.parametrize("convention", ["TDSE", "LvN"])
def test_liouvillian_generator_sparse_and_matrix_free(ctx, convention):
    """liouvillian((H0, (H1, eps)), c_ops; convention) (src/generators.jl:515-631): the sparse
    superoperator generator (dissipator folded into the drift, the commutator of H1 driven by
    eps) and the matrix-free one give the same Newton propagation; with :TDSE it is the
    Lindblad dynamics (trace 1, Hermitian rho)."""
    rng = np.random.default_rng(5)
    n = 12
    H0 = synth.dense_hermitian(n, rho=2.0, rng=rng)
    H1 = synth.dense_hermitian(n, rho=0.8, rng=rng)
    cops = [0.3 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)]
    eps = lambda t: 0.5 * np.sin(2 * t)       # noqa: E731
    tlist = np.linspace(0, 1.0, 21)
    psi = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    psi /= np.linalg.norm(psi)
    rho0 = np.ascontiguousarray(np.outer(psi, psi.conj()).T).reshape(-1)
    Lsp = P.liouvillian((H0, (H1, eps)), cops, convention=convention)
    Lmf = P.liouvillian((H0, (H1, eps)), cops, convention=convention, matrix_free=True)
    assert isinstance(Lsp, P.Generator) and len(Lsp.ops) == 2 and len(Lsp.amplitudes) == 1
    func = None if convention == "TDSE" else "exp"          # LvN: d rho/dt = L rho  ->  exp(L dt)
    dt_sign = {}
    out_sp = P.propagate(rho0, Lsp, tlist, method="newton", ctx=ctx, m_max=8, func=func, **dt_sign)
    out_mf = P.propagate(rho0, Lmf, tlist, method="newton", ctx=ctx, m_max=8, func=func, **dt_sign)
    assert np.linalg.norm(out_sp - out_mf) < 1e-11
    if convention == "TDSE":
        rho = out_mf.reshape(n, n).T
        assert abs(np.trace(rho) - 1) < 1e-10 and np.linalg.norm(rho - rho.conj().T) < 1e-10
        ref = qo.propagate(rho0, qo.Generator([Lsp.ops[0], Lsp.ops[1]], [eps]), tlist, "newton", m_max=8)
        assert np.linalg.norm(out_mf - ref) < 1e-10
    with pytest.raises(ValueError):
        P.liouvillian(H0, cops, convention="foo")
    with pytest.raises(ValueError):
        P.liouvillian(None, (), convention="TDSE")


def test_liouvillian_reference_tests(ctx):
    """test/test_liouvillian.jl restated.  "TLS dissipation" (:12-50): pure dissipator, analytic
    rho(T) -- propagated here with Newton (the reference uses :expprop) through the sparse and the
    matrix-free operator.  "LvN" (:53-124): N = 100, H = H0 + eps H1, 99 decay + 100 dephasing
    Lindblad operators; L rho equals the Lindblad right-hand side for both conventions, with and
    without the control term, for the sparse generator and the matrix-free operator."""
    ket = lambda i, n: np.eye(n, dtype=complex)[:, i]                     # noqa: E731
    ketbra = lambda i, j, n: np.outer(ket(i, n), ket(j, n).conj())        # noqa: E731
    g1, g2, T = 0.5, 0.2, 1.0
    A1, A2 = np.sqrt(g1) * ketbra(0, 1, 2), np.sqrt(2 * g2) * ketbra(1, 1, 2)
    psi0 = (ket(0, 2) + ket(1, 2)) / np.sqrt(2)
    rho0 = np.ascontiguousarray(np.outer(psi0, psi0.conj()).T).reshape(-1)
    expected = 0.5 * np.array([[2 - np.exp(-g1 * T), np.exp(-(g1 / 2 + g2) * T)],
                               [np.exp(-(g1 / 2 + g2) * T), np.exp(-g1 * T)]], dtype=complex)
    tlist = np.linspace(0, T, 11)
    for mf in (False, True):
        Lg = P.liouvillian(None, [A1, A2], convention="TDSE", matrix_free=mf)
        out = P.propagate(rho0, Lg, tlist, method="newton", ctx=ctx, m_max=3)
        rho = out.reshape(2, 2).T
        assert abs(1 - np.trace(rho)) < 1e-12 and abs(np.trace(rho @ rho)) < 1.0
        assert np.linalg.norm(rho - expected) < 1e-11
    # ---- "LvN"
    rng = np.random.default_rng(100)
    N = 100
    H0 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=0.1, rng=rng)
    Hfull = H0 + H1
    psi = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi /= np.linalg.norm(psi)
    r0 = np.outer(psi, psi.conj())
    vec = np.ascontiguousarray(r0.T).reshape(-1)
    unvec = lambda v: v.reshape(N, N).T                                     # noqa: E731
    x = L.State(ctx, data=vec)

    def apply(gen_or_op):
        if isinstance(gen_or_op, L.Operator):
            op = gen_or_op
        else:
            op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, sp.csr_matrix(gen_or_op))])
        y = L.State(ctx, n=N * N)
        op.mul(x, y)
        return unvec(y.numpy())
    comm = Hfull @ r0 - r0 @ Hfull
    assert np.linalg.norm(1j * comm - apply(P.liouvillian(Hfull, convention="LvN"))) < 1e-13
    assert np.linalg.norm(comm - apply(P.liouvillian(Hfull, convention="TDSE"))) < 1e-13
    cops = [np.sqrt(0.2) * ketbra(0, i, N) for i in range(1, N)] + [np.sqrt(0.1) * ketbra(i, i, N) for i in range(N)]
    diss = sum(A @ r0 @ A.conj().T - 0.5 * (A.conj().T @ A @ r0) - 0.5 * (r0 @ A.conj().T @ A) for A in cops)
    Lm = P.liouvillian(H0, [sp.csr_matrix(A) for A in cops], convention="LvN")
    assert sp.issparse(Lm)
    assert np.linalg.norm(apply(Lm) - (1j * (H0 @ r0 - r0 @ H0) + diss)) < 1e-13
    Lg = P.liouvillian((H0, (H1, lambda t: 1.0)), [sp.csr_matrix(A) for A in cops], convention="LvN")
    assert isinstance(Lg, P.Generator) and len(Lg.ops) == 2
    assert np.linalg.norm(apply(Lg.ops[0] + Lg.ops[1] * Lg.amplitudes[0](0.0)) - (1j * comm + diss)) < 1e-13
    Lmf = L.Liouvillian(ctx, [H0, H1], cops, ncoeffs=1, convention="LvN")      # 199 Lindblad operators: library GEMM chain
    Lmf.set_coeffs([1.0])
    assert np.linalg.norm(apply(Lmf) - (1j * comm + diss)) < 1e-12


def test_propagate_steps_error_behaviour(ctx):
    """qp_propagate keeps cheby!'s argument checks (src/cheby.jl:157) on both of its paths
    (persistent small-system kernel and launch-per-term loop) and rejects malformed calls."""
    rng = np.random.default_rng(3)
    for N in (12, 3000):              # register-resident / general
        H = synth.sparse_random(N, min(1.0, 6.0 / N), rho=2.0, hermitian=True, rng=rng)
        Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, H)])
        psi = L.State(ctx, data=synth.random_state(N))
        wrk = L.ChebyWrk(ctx, N, 8.0, -4.0, 0.1)
        with pytest.raises(L.QPAssertionError, match="wrk was initialized for dt"):
            L.propagate_steps(Op, psi, wrk, np.array([0.1, 0.2]))
        if N == 12:                   # one launch for the whole grid: the direction cannot change inside it
            with pytest.raises(L.QPArgumentError, match="change sign"):
                L.propagate_steps(Op, psi, wrk, np.array([0.1, -0.1]))
        wrong = L.Operator(ctx, [L.Matrix.from_scipy(ctx, sp.identity(N + 1, dtype=complex, format="csr"))])
        with pytest.raises(L.QPArgumentError, match="wrong shape"):
            L.propagate_steps(Op, psi, wrk, np.array([0.1]), observables=[wrong])
        with pytest.raises(L.QPArgumentError):
            L.propagate_steps(Op, L.State(ctx, n=N + 1), wrk, np.array([0.1]))
        # zero steps: the initial row only
        ev, st = L.propagate_steps(Op, psi, wrk, np.zeros(0), observables=[Op], store_states=True)
        assert ev.shape == (1, 1) and st.shape == (1, N)
        assert np.array_equal(st[0], psi.numpy())


def test_time_dependent_observables(ctx):
    """Observables as functions of ``(state, tlist, i)`` besides functions of the state and
    matrices (map_observable, src/storage.jl:100-123; test/test_timedependent_observables.jl):
    a rotating-frame population evaluated with the time of the storage slot."""
    rng = np.random.default_rng(8)
    N = 20
    H = synth.dense_hermitian(N, rho=2.0, rng=rng)
    tlist = np.linspace(0, 1.0, 9)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    target = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    target /= np.linalg.norm(target)
    O = synth.dense_hermitian(N, rho=1.0, rng=rng)
    rotating = lambda psi, tl, i: np.exp(1j * 0.7 * tl[i]) * np.vdot(target, psi)        # noqa: E731
    overlap = lambda psi: abs(np.vdot(target, psi)) ** 2                                  # noqa: E731
    for backward in (False, True):
        _, ev = P.propagate(psi0, (H,), tlist, method="cheby", storage=True, backward=backward, ctx=ctx,
                            observables=[rotating, overlap, O], E_min=-4.0, E_max=4.0)
        _, st = P.propagate(psi0, (H,), tlist, method="cheby", storage=True, backward=backward, ctx=ctx,
                            E_min=-4.0, E_max=4.0)
        for i in range(len(tlist)):
            assert abs(ev[0, i] - np.exp(1j * 0.7 * tlist[i]) * np.vdot(target, st[:, i])) < 1e-12
            assert abs(ev[1, i] - abs(np.vdot(target, st[:, i])) ** 2) < 1e-12
            assert abs(ev[2, i] - np.vdot(st[:, i], O @ st[:, i])) < 1e-12


def test_propagator_property_access(ctx):
    """test/test_prop_interfaces.jl:309-336 ("Propagator property access"): generator hidden,
    state / tlist / t not assignable, parameters re-bindable, the public property names."""
    rng = np.random.default_rng(6)
    N = 10
    tlist = np.linspace(0, 10, 101)
    H0 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=0.2, rng=rng)
    psi = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi /= np.linalg.norm(psi)
    p = P.init_prop(psi, P.hamiltonian(H0, (H1, lambda t: np.sin(t))), tlist, "cheby", ctx=ctx, E_min=-3.0, E_max=3.0)
    with pytest.raises(AttributeError):
        p.generator
    for name, value in (("state", p.state), ("tlist", tlist), ("t", 0.0), ("generator", H0)):
        with pytest.raises(AttributeError):
            setattr(p, name, value)
    new_params = [par.copy() for par in p.parameters]
    p.parameters = new_params
    assert p.parameters is new_params
    assert p.propertynames() == ("state", "tlist", "t", "parameters", "backward", "inplace")
    priv = p.propertynames(True)
    assert len(priv) >= 6 and all(q in priv for q in p.propertynames())
    P.prop_step(p)                       # the propagator itself still advances state and t
    assert p.t == tlist[1]
    with pytest.raises(ValueError):      # "init_prop unknown method" (:371-381)
        P.init_prop(psi, H0, tlist, 42, ctx=ctx)


def test_ode_function(ctx):
    """src/ode_function.jl:54-98: f(du, u, p, t) = c H(t) u on the device, in place and not in
    place, with `vals_dict` substitution; integrating it with classical RK4 reproduces the
    Chebyshev propagation of the same generator."""
    H0, psi0 = _optomech()
    N = H0.shape[0]
    H1 = sp.diags([np.linspace(-1, 1, N).astype(complex)], [0], format="csr")
    eps = lambda t: 0.3 * np.sin(20.0 * t)  # noqa: E731
    gen = P.hamiltonian(H0, (H1, eps))
    tlist = np.linspace(0, 0.2, 2001)
    f = P.ode_function(gen, tlist, ctx=ctx)
    u = L.State(ctx, data=psi0)
    du = L.State(ctx, n=N)
    out = f(du, u, None, 0.2)
    assert out is du
    ref = -1j * ((H0 + eps(0.2) * H1) @ psi0)
    assert np.linalg.norm(du.numpy() - ref) < 1e-12
    new = f(u, {}, 0.2)                                     # not in place: a new state
    assert new is not u and np.linalg.norm(new.numpy() - ref) < 1e-12
    sub = f(u, [(eps, 2.0)], 0.2)                           # vals_dict, keyed by identity
    assert np.linalg.norm(sub.numpy() - (-1j * ((H0 + 2.0 * H1) @ psi0))) < 1e-12
    g = P.ode_function(P.hamiltonian(H0, (H1, np.zeros(len(tlist)))), tlist, ctx=ctx, c=1.0)
    with pytest.raises(TypeError, match="is invalid"):      # src/controls.jl:388-396
        g(u, None, 0.2)
    arr = g.generator.amplitudes[0]
    assert np.linalg.norm(g(u, [(arr, 0.5)], 0.2).numpy() - ((H0 + 0.5 * H1) @ psi0)) < 1e-12
    # RK4 over the grid against the PWC Chebyshev propagation (midpoint values) of the same H(t)
    k = [L.State(ctx, n=N) for _ in range(4)]
    tmp = L.State(ctx, n=N)
    for i in range(len(tlist) - 1):
        t, h = tlist[i], tlist[i + 1] - tlist[i]
        f(k[0], u, None, t)
        tmp.copy_from(u); tmp.axpy(h / 2, k[0])             # noqa: E702
        f(k[1], tmp, None, t + h / 2)
        tmp.copy_from(u); tmp.axpy(h / 2, k[1])             # noqa: E702
        f(k[2], tmp, None, t + h / 2)
        tmp.copy_from(u); tmp.axpy(h, k[2])                 # noqa: E702
        f(k[3], tmp, None, t + h)
        for w, kk in zip((h / 6, h / 3, h / 3, h / 6), k):
            u.axpy(w, kk)
    cheb = P.propagate(psi0, gen, tlist, method="cheby", ctx=ctx)
    cheb = cheb.numpy() if hasattr(cheb, "numpy") else np.asarray(cheb)
    assert np.linalg.norm(u.numpy() - cheb) < 1e-6          # O(dt^2) PWC vs O(dt^4) RK4 on the same H(t)
    assert abs(np.linalg.norm(u.numpy()) - 1.0) < 1e-8


def test_matvec_counters_like_reference_timings(ctx):
    """test/test_timings.jl:8-39 with the context's counters in the place of TimerOutputs: N = 10, 100 steps
    of a time-dependent generator with Cheby -> more than 200 "matrix-vector product" calls.  And the
    qualitative statement of docs/src/benchmarks/profiling.md:112 (N = 200, 100 steps of dt = 1, spectral
    radius 1): the Newton propagator (m_max = 10, chunks of 10 applications) needs more applications of
    H than Chebychev, whose count per step is its number of coefficients minus one."""
    rng = np.random.default_rng(677918056 % 2**32)
    N = 10
    tlist = np.linspace(0, 10, 101)
    H0 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    H1 = synth.dense_hermitian(N, rho=0.1, rng=rng)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    p = P.init_prop(psi0, P.hamiltonian(H0, (H1, lambda t: np.sin(t))), tlist, method="cheby", ctx=ctx,
                    rng=np.random.default_rng(5))
    ctx.sync()
    ctx.reset_stats()
    for _ in range(len(tlist) - 1):
        P.prop_step(p)
    st = ctx.stats()
    assert st["n_cheby_steps"] == 100
    assert st["n_matvec"] > 200
    assert st["n_matvec"] == 100 * (len(p.wrk.coeffs) - 1)

    N = 200
    tlist = np.arange(0, 101, 1.0)
    H0 = synth.dense_hermitian(N, rho=1.0, rng=rng)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    counts, outs = {}, {}
    for method in ("cheby", "newton"):
        ctx.sync()
        ctx.reset_stats()
        outs[method] = P.propagate(psi0, (H0,), tlist, method=method, ctx=ctx, rng=np.random.default_rng(6))
        counts[method] = ctx.stats()["n_matvec"]
    assert np.linalg.norm(outs["cheby"] - outs["newton"]) < 1e-10
    assert counts["newton"] > counts["cheby"] > 200
    assert counts["newton"] >= 100 * 20 and counts["newton"] % 10 == 0   # whole chunks of m_max = 10, at least one restart


def test_named_profiler_ranges_do_not_disturb_a_step(ctx):
    """Knob roctx = 1 (or QP_ROCTX=1): the steps' phases are wrapped in named ranges carrying the reference's TimerOutputs
    section names ("prop_step!", "matrix-vector product", "arnoldi!", "diagonalize_hessenberg_matrix", "get Leja points",
    "get Newton coeffs", "evaluate polynomial": test/test_timings.jl:28-30, src/newton.jl:276-328) for rocprofv3
    --marker-trace.  The marker library is optional (resolved with dlopen); results are the same bits with it on."""
    import qprop_amd.synth as synth
    Lm = synth.liouvillian_tridiag(40)
    N = Lm.shape[0]
    rho0 = synth.random_state(N)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lm)])
    outs = []
    for on in (0, 1):
        ctx.tuning_set("roctx", on)
        try:
            rho = L.State(ctx, data=rho0)
            L.newton(rho, op, 0.4, L.NewtonWrk(ctx, N, m_max=10))
            rp, col, vals = synth.hermitian_offsets_csr(4096, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
            oph = L.Operator(ctx, [L.Matrix(ctx, 4096, 4096, rp, col, vals)])
            psi = L.State(ctx, data=synth.random_state(4096))
            L.cheby(psi, oph, 1.0, L.ChebyWrk(ctx, 4096, 20.0, -10.0, 1.0))
            outs.append((rho.numpy(), psi.numpy()))
        finally:
            ctx.tuning_set("roctx", 0)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
