"""CPU-side checks of the C ABI: the library loads, exports every symbol that
include/qprop.h declares, its host-only numerics (Bessel coefficients, Hessenberg
eigenvalues, Leja ordering, Newton divided differences) agree with the oracle, its
index work is bit-exact, and the product path fails loudly without a GPU.
No device compute happens here."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "qprop.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(qp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"qp_func_cb"}
    assert len(declared) > 50
    lib = L.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"symbols declared in qprop.h but not exported: {missing}"
    unbound = sorted(declared - set(L.SIGNATURES))
    assert not unbound, f"symbols without a ctypes signature: {unbound}"
    assert lib.qp_version() >= 100


def test_no_gpu_fails_loudly():
    """The product path has no CPU fallback."""
    if L.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(L.QPError) as ei:
        L.Context(0)
    assert ei.value.status == 8 and "no CPU fallback" in str(ei.value)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "quantumpropagators.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "qp_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


@pytest.mark.parametrize("Delta,dt", [(20.0, 1.0), (849.53, 0.5), (4.0, 1.0), (100.0, 1.0), (20.2, 0.05)])
def test_cheby_coeffs_match_oracle(Delta, dt):
    """src/cheby.jl:25-39; glibc jn vs scipy jv.  Count within +-1 (the append-then-test
    at the 1e-12 limit), values to 5e-14 absolute (two libm-grade Bessel implementations at alpha ~ 212)."""
    a = L.cheby_coeffs(Delta, dt)
    b = qo.cheby_coeffs(Delta, dt)
    assert abs(len(a) - len(b)) <= 1
    n = min(len(a), len(b))
    assert np.max(np.abs(a[:n] - b[:n])) < 5e-14
    assert abs(a[-1]) <= 1e-12 < abs(a[-2])


def test_cheby_coeffs_reference_count():
    """test/test_cheby.jl:36: Delta = spectral range of Hermitian(rand(ComplexF64,N,N)),
    N=1000 (~849.5), dt=0.5 gives 267 or 268 coefficients; BASELINE config C2
    (alpha=10) gives 32."""
    assert len(L.cheby_coeffs(20.0, 1.0)) == 32
    assert len(L.cheby_coeffs(849.53, 0.5)) in (267, 268)


@pytest.mark.parametrize("m,herm", [(1, False), (2, False), (3, False), (5, True), (20, False), (60, True), (200, False)])
def test_hessenberg_eigvals(m, herm):
    """src/arnoldi.jl:143-170 with our own complex QR standing in for LAPACK."""
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m + 1, m + 1)) + 1j * rng.standard_normal((m + 1, m + 1))
    if herm:
        A = (A + A.conj().T) / 2
    Hs = np.asfortranarray(np.triu(A, -1))
    ev = L.hessenberg_eigvals(Hs, m)
    ref = qo.diagonalize_hessenberg_matrix(Hs, m)
    assert np.max(np.abs(np.sort_complex(ev) - np.sort_complex(ref))) < 1e-11 * max(1.0, np.max(np.abs(ref)))
    if m > 2:  # sorted by (real, imag)
        assert np.all(np.diff(ev.real) >= 0)
    if m <= 20:
        acc = L.hessenberg_eigvals(Hs, m, accumulate=True)
        ref = qo.diagonalize_hessenberg_matrix(Hs, m, accumulate=True)
        assert len(acc) == m * (m + 1) // 2
        off = 0
        for j in range(1, m + 1):
            assert np.max(np.abs(np.sort_complex(acc[off:off + j]) - np.sort_complex(ref[off:off + j]))) < 1e-11
            off += j


def test_hessenberg_eigvals_from_arnoldi():
    rng = np.random.default_rng(3)
    H = synth.dense_nonhermitian(300, rng=rng)
    psi = rng.standard_normal(300) + 1j * rng.standard_normal(300)
    psi /= np.linalg.norm(psi)
    m = 30
    Hess = np.zeros((m + 1, m + 1), dtype=complex, order="F")
    q = [np.empty(300, dtype=complex) for _ in range(m + 1)]
    qo.arnoldi(Hess, q, m, psi, H, 0.5)
    ev = L.hessenberg_eigvals(Hess, m, accumulate=True)
    ref = qo.diagonalize_hessenberg_matrix(Hess, m, accumulate=True)
    off = 0
    for j in range(1, m + 1):
        assert np.max(np.abs(np.sort_complex(ev[off:off + j]) - np.sort_complex(ref[off:off + j]))) < 1e-12
        off += j


def test_extend_leja_matches_oracle():
    """src/newton.jl:97-148 -- selection order is index work: identical picks."""
    rng = np.random.default_rng(5)
    leja_a = np.zeros(41, dtype=complex)
    leja_b = leja_a.copy()
    n_a = n_b = 0
    for rnd in range(4):
        m = 10
        pts = rng.standard_normal(m * (m + 1) // 2) + 1j * rng.standard_normal(m * (m + 1) // 2)
        leja_a, n_a = qo.extend_leja(leja_a, n_a, pts.copy(), m)
        leja_b, n_b = L.extend_leja(leja_b, n_b, pts.copy(), m)
        assert n_a == n_b == (rnd + 1) * m
        assert np.array_equal(leja_a[:n_a], leja_b[:n_b])


@pytest.mark.parametrize("case", ["duplicates", "near_ties", "nested_ritz", "tiny_and_huge", "on_leja_points"])
def test_extend_leja_adversarial_ties_match_oracle(case):
    """The Leja order is index work (bit-exact): the library ranks candidates by an exactly scaled product and lets the
    reference's own hypot / pow chain (src/newton.jl:132-136) decide among candidates within its rounding noise.
    Adversarial inputs -- exact duplicates, candidates 1e-15 apart, the accumulated Ritz values of nested Hessenberg
    blocks of a Hermitian matrix whose extremal eigenvalues have converged (what newton! actually feeds it), distances
    of 1e-170 and 1e+170 (the squared distance alone would leave the double range), candidates that coincide with
    existing Leja points (product zero) -- give the oracle's picks, index for index, over several restarts."""
    rng = np.random.default_rng(11)
    m = 12
    ncand = m * (m + 1) // 2

    def candidates(rnd):
        if case == "duplicates":
            base = rng.standard_normal(ncand // 3 + 1) + 1j * rng.standard_normal(ncand // 3 + 1)
            return np.resize(np.repeat(base, 3), ncand).copy()
        if case == "near_ties":
            base = rng.standard_normal(ncand // 4 + 1) + 1j * rng.standard_normal(ncand // 4 + 1)
            pts = np.resize(np.repeat(base, 4), ncand).copy()
            return pts * (1.0 + 1e-15 * rng.integers(-4, 5, ncand)) + 1e-16 * rng.integers(-3, 4, ncand)
        if case == "nested_ritz":
            A = rng.standard_normal((60, 60))
            A = (A + A.T) / 2 + np.diag(np.linspace(-30, 30, 60))       # well separated extremal eigenvalues: fast convergence
            v = rng.standard_normal(60) + 0j
            Hs = np.zeros((m + 1, m + 1), dtype=complex)
            qv = [np.zeros(60, dtype=complex) for _ in range(m + 1)]
            mm = qo.arnoldi(Hs, qv, m, v / np.linalg.norm(v), A.astype(complex), 1.0, extended=True)
            assert mm == m
            return np.array(qo.diagonalize_hessenberg_matrix(Hs, mm, accumulate=True))[:ncand].copy()
        if case == "tiny_and_huge":
            scale = np.array([1e-170, 1e-90, 1.0, 1e90, 1e170])[rng.integers(0, 5, ncand)]
            return (rng.standard_normal(ncand) + 1j * rng.standard_normal(ncand)) * scale
        pts = rng.standard_normal(ncand) + 1j * rng.standard_normal(ncand)
        if rnd > 0:
            pts[::3] = leja_a[rng.integers(0, n_a, len(pts[::3]))]           # candidates ON earlier Leja points
        return pts

    leja_a = np.zeros(4 * m + 1, dtype=complex)
    leja_b = leja_a.copy()
    n_a = n_b = 0
    for rnd in range(4):
        pts = candidates(rnd)
        assert len(pts) >= m
        with np.errstate(over="ignore", under="ignore", divide="ignore", invalid="ignore"):
            leja_a, n_a = qo.extend_leja(leja_a, n_a, pts.copy(), m)
        leja_b, n_b = L.extend_leja(leja_b, n_b, pts.copy(), m)
        assert n_a == n_b == (rnd + 1) * m
        assert np.array_equal(leja_a[:n_a], leja_b[:n_b]), (case, rnd)


@pytest.mark.parametrize("func", ["expmi", "exp", "callback"])
def test_extend_newton_coeffs_matches_oracle(func):
    """src/newton.jl:176-214."""
    rng = np.random.default_rng(6)
    pyf = {"expmi": lambda z: np.exp(-1j * z), "exp": np.exp, "callback": lambda z: 1.0 / (2.5 + z)}[func]
    arg = func if func != "callback" else pyf
    leja = np.zeros(0, dtype=complex)
    a1 = np.zeros(31, dtype=complex)
    a2 = a1.copy()
    n1 = n2 = 0
    nl = 0
    for rnd in range(3):
        m = 8
        pts = (rng.standard_normal(36) + 1j * rng.standard_normal(36)) * 0.8
        leja, nl = qo.extend_leja(leja, nl, pts, m)
        a1, n1 = qo.extend_newton_coeffs(a1, n1, leja, pyf, nl, 2.0)
        a2, n2 = L.extend_newton_coeffs(a2, n2, leja, arg, nl, 2.0)
        assert n1 == n2 == nl
        # divided differences cancel: compare absolutely, relative to the largest coefficient
        assert np.max(np.abs(a1[:n1] - a2[:n2])) < 1e-12 * np.max(np.abs(a1[:n1]))


def test_newton_coeffs_underflow_status():
    leja = np.array([1.0, 1.0 + 1e-210], dtype=complex)
    a = np.zeros(4, dtype=complex)
    with pytest.raises(L.QPAssertionError, match="Divided differences too small"):
        L.extend_newton_coeffs(a, 0, leja, "exp", 2, 1.0)


def test_csc_to_csr_bit_exact():
    rng = np.random.default_rng(7)
    for n, dens in ((1, 1.0), (17, 0.3), (130, 0.05), (64, 0.0)):
        A = synth.sparse_random(n, dens, rng=rng).tocsc() if dens > 0 else __import__("scipy.sparse").sparse.csc_matrix((n, n), dtype=complex)
        A.sort_indices()
        got = L.csc_to_csr(n, n, A.indptr + 1, A.indices + 1, A.data, index_base=1)
        ref = qo.csc_to_csr(n, n, A.indptr + 1, A.indices + 1, A.data, index_base=1)
        for g, r in zip(got, ref):
            assert g.dtype == r.dtype and np.array_equal(g, r)


def test_csc_pointer_array_is_validated():
    """The boundary rejects a pointer array before indexing through it: colptr must start at the index
    base, be monotone and stay within nnz (a bad one used to index rowval / nzval out of bounds)."""
    import pytest
    n = 6
    colptr = np.array([1, 3, 3, 5, 6, 6, 7], dtype=np.int64)       # 1-based, 6 entries
    rowval = np.array([1, 4, 2, 6, 3, 5], dtype=np.int64)
    nzval = np.arange(6, dtype=np.complex128)
    L.csc_to_csr(n, n, colptr, rowval, nzval, index_base=1)         # the good one passes
    for bad in (np.array([2, 3, 3, 5, 6, 6, 7]),                    # colptr[0] != base
                np.array([1, 3, 2, 5, 6, 6, 7]),                    # not monotone
                np.array([1, 3, 3, 9, 6, 6, 7]),                    # an entry beyond nnz
                np.array([1, 3, 3, 5, 6, 6, 0])):                   # negative nnz
        with pytest.raises(L.QPArgumentError):
            L.csc_to_csr(n, n, bad.astype(np.int64), rowval, nzval, index_base=1)
    with pytest.raises(L.QPArgumentError):                          # row index out of range
        L.csc_to_csr(n, n, colptr, rowval + 1, nzval, index_base=1)


def test_partition_rows_bit_exact():
    rng = np.random.default_rng(8)
    lens = rng.integers(0, 9, 1000)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for parts in (1, 2, 3, 8):
        for bal in ("rows", "nnz"):
            assert np.array_equal(L.partition_rows(rowptr, parts, bal), qo.partition_rows(rowptr, parts, bal))


@pytest.mark.parametrize("nx,ny,nnn", [(100, 60, False), (64, 90, False), (130, 50, True), (70, 400, False)])
def test_lattice_fill_host_completes_open_boundary_grids(nx, ny, nnn):
    """Index work of the lattice completion (engine_plans.hip: lattice_fill, the host half of qp_operator_create), bit-exact
    against a NumPy restatement: every row between the first and the last grid row of an open-boundary grid Hamiltonian
    gets the full distance list of the stencil (explicit zeros where the grid's x-edge cut a neighbour off), the entries
    that land in the first / last grid row get their transposes, nothing else changes."""
    H = synth.grid_hamiltonian_2d(nx, ny, flux=0.3, next_nearest=nnn)
    N = nx * ny
    rp, col = L.lattice_fill_host(N, N, H.indptr, H.indices, min_blocks=16)
    D = [-nx, -2, -1, 0, 1, 2, nx] if nnn else [-nx, -1, 0, 1, nx]
    want = set(zip(*H.nonzero()))
    want |= {(r, r) for r in range(N)}
    for r in range(nx, N - nx):
        for d in D:
            want.add((r, r + d))
            want.add((r + d, r))
    got = {(r, int(c)) for r in range(N) for c in col[rp[r]:rp[r + 1]]}
    assert got == want
    for r in range(N):                              # sorted, unique
        assert np.all(np.diff(col[rp[r]:rp[r + 1]]) > 0)
    assert rp[-1] == len(col) == len(want)


def test_lattice_fill_host_thin_three_dimensional_grid_needs_ten_percent_more_entries():
    """A 64 x 4 x 1024 seven-point grid with open boundaries: its y-direction is four points wide, so 9.7 % of the completed
    pattern is explicit zeros -- inside the 12 % the completion accepts and more than the 5 % the wrapper used to leave room
    for (QP_E_BAD_ARG "col_out holds 1506854 entries, 1572210 needed")."""
    nx, ny, nz = 64, 4, 1024
    H = synth.grid_hamiltonian_3d(nx, ny, nz, flux=0.1)
    N = nx * ny * nz
    rp, col = L.lattice_fill_host(N, N, H.indptr, H.indices, min_blocks=16)
    added = int(rp[-1]) - H.nnz
    assert 0.08 * H.nnz < added < 0.12 * H.nnz
    mid = N // 2
    assert sorted(int(c) - mid for c in col[rp[mid]:rp[mid + 1]]) == [-nx * ny, -nx, -1, 0, 1, nx, nx * ny]


def test_lattice_fill_host_leaves_other_patterns_alone():
    """No completion for: a lattice with an entry outside its distances, more than 12 % missing entries, operators below the
    size knob, a complete lattice, a non-symmetric distance list, a row-partitioned pattern whose interior rows reach the
    halo columns; the local rows of a partitioned grid operator (halo columns only in the first / last grid row) are completed."""
    import scipy.sparse as sp
    nx, ny = 100, 60
    H = synth.grid_hamiltonian_2d(nx, ny)
    N = nx * ny

    def same(M, n=N, ncols=N, **kw):
        rp, col = L.lattice_fill_host(n, ncols, M.indptr, M.indices, **{"min_blocks": 16, **kw})
        return np.array_equal(rp, M.indptr) and np.array_equal(col, M.indices)

    assert not same(H)
    assert same(H, min_blocks=N // 64 + 1)                                      # below the size knob
    H2 = sp.lil_matrix(H)
    H2[N // 2 + 5, N // 2 + 37] = H2[N // 2 + 37, N // 2 + 5] = 0.5            # a foreign entry
    assert same(sp.csr_matrix(H2))
    Hh = sp.lil_matrix(H)
    for y in range(2, ny - 2):                                                  # every second x-hop cut: 20 % missing
        for x in range(0, nx - 1, 2):
            Hh[x + nx * y, x + 1 + nx * y] = 0
            Hh[x + 1 + nx * y, x + nx * y] = 0
    Hh = sp.csr_matrix(Hh)
    Hh.eliminate_zeros()
    assert same(Hh)
    rpb, colb, valb = synth.hermitian_offsets_csr(1 << 13, offsets=(1, 2, 128, 256))   # already complete (periodic wrap rows aside)
    assert same(synth.to_scipy(rpb, colb, valb, 1 << 13), n=1 << 13, ncols=1 << 13)
    U = sp.csr_matrix(sp.triu(H))                                              # distances not symmetric
    assert same(U)
    # local rows [nx * 10, nx * 50) of the grid with the halo columns appended behind the local ones
    lo, hi = nx * 10, nx * 50
    Hl = H[lo:hi].tocoo()
    ccol = np.where((Hl.col >= lo) & (Hl.col < hi), Hl.col - lo, np.where(Hl.col < lo, (hi - lo) + (Hl.col - (lo - nx)), (hi - lo) + nx + (Hl.col - hi)))
    Hloc = sp.csr_matrix((Hl.data, (Hl.row, ccol)), shape=(hi - lo, hi - lo + 2 * nx))
    Hloc.sort_indices()
    rp, col = L.lattice_fill_host(hi - lo, hi - lo + 2 * nx, Hloc.indptr, Hloc.indices, min_blocks=16)
    assert rp[-1] == Hloc.nnz + 2 * (40 - 2) + 2
    assert np.all(np.diff(rp)[nx:hi - lo - nx] == 5)


def test_accumulator_schedule_host():
    """qp_acc_schedule_host (include/qprop.h, qp_acc_defer): the Psi accumulator is touched
    every third term, the last term always, the first update no later than term 2 (Psi = v_0 is
    overwritten by term 2) -- and replaying the folded updates reproduces the term-by-term
    axpy sequence of src/cheby.jl:172/:182/:205 exactly."""
    import qprop_amd.lib as L
    rng = np.random.default_rng(0)
    for n_coeffs in range(2, 40):
        a = rng.standard_normal(n_coeffs)
        sched = L.acc_schedule(a)
        nterms = n_coeffs - 1
        assert len(sched) == nterms and not sched[nterms - 1].skip
        upd = [m for m in range(1, nterms + 1) if not sched[m - 1].skip]
        assert upd[0] <= 2 and all(b - c <= 3 for b, c in zip(upd[1:], upd[:-1]))
        assert len(upd) <= nterms // 3 + 2
        # v_0 .. v_nterms as random numbers; sequential reference
        v = rng.standard_normal(nterms + 1) + 1j * rng.standard_normal(nterms + 1)
        ref = a[0] * v[0]
        for m in range(1, nterms + 1):
            ref = ref + a[m] * v[m]
        acc, last = None, 0
        for m in range(1, nterms + 1):
            d = sched[m - 1]
            if d.skip:
                continue
            assert d.n_defer == m - last - 1
            r = acc if acc is not None else a[0] * (v[m - 2] if d.n_defer == 1 else v[m - 1])
            if acc is None:
                assert (m - 1 - d.n_defer) == 0          # the first update starts from Psi = v_0
            if d.n_defer == 2:
                r = r + d.a_d2 * v[m - 2]
            if d.n_defer >= 1:
                r = r + d.a_d1 * v[m - 1]
            acc = r + a[m] * v[m]
            last = m
        assert acc == ref


# ---- property tests of the boundary's index work (bit-exact, no GPU) ------------------------

from hypothesis import given, settings, strategies as st   # noqa: E402


@st.composite
def _sparse_csc(draw):
    """A Julia-style SparseMatrixCSC: 1-based Int64 colptr / rowval, rows ascending per column,
    possibly empty columns and rows, rectangular."""
    nrows = draw(st.integers(1, 40))
    ncols = draw(st.integers(1, 40))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    colptr = [1]
    rowval = []
    for _ in range(ncols):
        k = int(rng.integers(0, min(nrows, 7) + 1)) if rng.random() > 0.2 else 0
        rows = np.sort(rng.choice(nrows, size=k, replace=False)) + 1
        rowval.extend(int(r) for r in rows)
        colptr.append(colptr[-1] + k)
    nz = rng.standard_normal(len(rowval)) + 1j * rng.standard_normal(len(rowval))
    return nrows, ncols, np.array(colptr, dtype=np.int64), np.array(rowval, dtype=np.int64), nz


@settings(max_examples=150, deadline=None, derandomize=True)
@given(_sparse_csc())
def test_csc_to_csr_property(m):
    """qp_csc_to_csr_host against SciPy for arbitrary (empty rows/columns, rectangular) CSC input:
    same pattern, same values bit for bit, columns ascending within a row, and identical to the
    oracle's restatement."""
    import scipy.sparse as sp
    import qprop_amd.lib as L
    from oracle import qp_oracle as qo
    nrows, ncols, colptr, rowval, nz = m
    rp, col, vals = L.csc_to_csr(nrows, ncols, colptr, rowval, nz)
    A = sp.csc_matrix((nz, rowval - 1, colptr - 1), shape=(nrows, ncols)).tocsr()
    A.sort_indices()
    assert np.array_equal(rp, A.indptr) and np.array_equal(col, A.indices) and np.array_equal(vals, A.data)
    orp, ocol, ovals = qo.csc_to_csr(nrows, ncols, colptr, rowval, nz)
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and np.array_equal(vals, ovals)
    for r in range(nrows):
        assert np.all(np.diff(col[rp[r]:rp[r + 1]]) > 0)


@settings(max_examples=150, deadline=None, derandomize=True)
@given(st.lists(st.integers(0, 12), min_size=1, max_size=300), st.integers(1, 9), st.sampled_from(["rows", "nnz"]))
def test_partition_rows_property(lens, nparts, balance):
    """qp_partition_rows_host: contiguous blocks that tile [0, nrows), monotone, identical to the
    oracle, and balanced: by rows within one row, by nnz within the longest row."""
    import qprop_amd.lib as L
    from oracle import qp_oracle as qo
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nrows = len(lens)
    b = L.partition_rows(rp, nparts, balance)
    assert b[0] == 0 and b[-1] == nrows and np.all(np.diff(b) >= 0) and len(b) == nparts + 1
    assert np.array_equal(b, qo.partition_rows(rp, nparts, balance))
    if balance == "rows":
        sizes = np.diff(b)
        assert sizes.max() - sizes.min() <= 1
    else:
        w = rp[b[1:]] - rp[b[:-1]]
        assert w.max() <= rp[-1] / nparts + max(lens) + 1


def test_host_numerics_under_sanitizers(tmp_path):
    """The host numerics of the library, compiled with g++ -fsanitize=address,undefined and run on
    random inputs (tests/sanitize_host_numerics.cpp): no out-of-bounds access, no undefined behaviour."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    csrc = os.path.join(ROOT, "quantumpropagators.jl_amd", "csrc")
    exe = str(tmp_path / "host_san")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-I", csrc, os.path.join(csrc, "host_numerics.cpp"),
                            os.path.join(ROOT, "tests", "sanitize_host_numerics.cpp"), "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "sanitizer run clean" in run.stdout


def test_host_index_work_under_sanitizers(tmp_path):
    """the HOST INDEX WORK of operator creation -- format choice, lattice completion, strip-walk plan, column
    encodings, Hermitian packing, column-blocked mirror, value dictionary, the get_csr decoders -- compiled from the library's own
    sources (csrc/engine_core.hip, engine_plans.hip, host_numerics.cpp) with g++ -fsanitize=address,undefined against a host
    stand-in of the HIP runtime (tests/hip_host_shim: device memory = heap memory, so every copy into a "device" array is
    bounds-checked) and fuzzed over tall / wide / 1-row / N < 64 / empty-row shapes, lattices and spin chains
    (tests/sanitize_host_index.cpp; the 5681 x 358 operator of the round-4 GPU memory fault is the first case).  Every stored slot
    -- pads and the lanes beyond the last row included -- must decode to a column inside the matrix, every pad must be zero."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    csrc = os.path.join(ROOT, "quantumpropagators.jl_amd", "csrc")
    exe = str(tmp_path / "host_index_san")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-x", "c++",
                            "-I", os.path.join(ROOT, "tests", "hip_host_shim"), "-I", csrc,
                            os.path.join(ROOT, "tests", "sanitize_host_index.cpp"), os.path.join(csrc, "host_numerics.cpp"),
                            "-o", exe, "-lpthread"], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    # (the context record is deliberately never freed -- handles may be destroyed in any order, csrc/engine.h -- so the
    # leak checker is off; every other check is on)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "sanitizer run clean" in run.stdout
    n = {k: int(v) for v, k in __import__("re").findall(r"(\d+) (operators|Hermitian-packed|with a strip-walk plan|with a value dictionary|with a column-blocked mirror)", run.stdout)}
    assert n["operators"] > 300 and n["with a strip-walk plan"] > 20 and n["with a value dictionary"] > 20 and n["with a column-blocked mirror"] >= 4
