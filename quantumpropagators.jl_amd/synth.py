"""Synthetic inputs for the `prop_step!` hot path.

Counter-based (splitmix64) so that every rank / language regenerates identical
bits from (seed, row, k) without communicating.  Host-side NumPy only; nothing
here touches the GPU.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

BANDED_OFFSETS = (1, 2, 3, 4, 1024, 2048, 3072, 4096)
DEFAULT_SEED = 20260612


def splitmix64(x):
    """One splitmix64 output step on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + _GOLDEN).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _u01(h):
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def coupling(seed, i, k, rho, n_offsets):
    """g(seed, i, k) = rho/(2 n_offsets) * u * exp(2 pi i phi)."""
    i = np.asarray(i, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) ^ (i * _GOLDEN + np.uint64(k) * _M1)
    h1 = splitmix64(x)
    h2 = splitmix64(h1)
    u = _u01(h1)
    phi = _u01(h2)
    return (rho / (2.0 * n_offsets)) * u * np.exp(2j * np.pi * phi)


def scattered_offsets(N, n_offsets=8, seed=DEFAULT_SEED):
    """Seeded distinct offsets in [1, N/2)."""
    out = []
    k = 0
    while len(out) < n_offsets:
        h = int(splitmix64(np.array([seed + 7919 * k], dtype=np.uint64))[0])
        d = 1 + h % (N // 2 - 1)
        if d not in out:
            out.append(d)
        k += 1
    return tuple(sorted(out))


def _usable_cpus():
    """Cores this process may keep busy: the affinity mask, capped by the container's CPU quota (cgroup cpu.max).  A thread
    pool sized by the host's core count (256 on the GPU boxes, quota 16) gets the whole control group THROTTLED for the rest
    of the scheduler period -- up to 100 ms in which no thread of the process runs, the one that enqueues kernels included
    (profiles/r04/n22_outlier.txt)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _offsets_rows(N, offsets, rho, seed, a, b, col_out, val_out):
    """Rows [a, b) of hermitian_offsets_csr into col_out / val_out (views of (b - a) x 2 nk)."""
    nk = len(offsets)
    rows = np.arange(a, b, dtype=np.int64)
    cols = np.empty((b - a, 2 * nk), dtype=np.int64)
    vals = np.empty((b - a, 2 * nk), dtype=np.complex128)
    for k, d in enumerate(offsets):
        cols[:, 2 * k] = (rows + d) % N
        vals[:, 2 * k] = coupling(seed, rows, k, rho, nk)
        src = (rows - d) % N
        cols[:, 2 * k + 1] = src
        vals[:, 2 * k + 1] = np.conj(coupling(seed, src, k, rho, nk))
    # columns ascending within a row.  Between two consecutive rows of {d_k, N - d_k} no column wraps differently, so every
    # row of such a segment has the SAME order: sort one row per segment and apply its permutation (for the banded offsets
    # the segments are [0, 1), [1, 2) ... near the ends and one long one in the middle; for scattered offsets up to N / 2
    # there are at most 4 len(offsets) + 1 of them)
    cuts = sorted({a, b} | {c for d in offsets for c in (d, N - d) if a < c < b})
    for s, e in zip(cuts[:-1], cuts[1:]):
        perm = np.argsort(cols[s - a], kind="stable")
        col_out[s - a:e - a] = cols[s - a:e - a][:, perm]
        val_out[s - a:e - a] = vals[s - a:e - a][:, perm]


def hermitian_offsets_csr(N, offsets=BANDED_OFFSETS, rho=10.0, seed=DEFAULT_SEED,
                          row_begin=0, row_end=None):
    """Hermitian H with exactly 2*len(offsets) nnz per row, no diagonal:
    H[i,(i+d_k)%N] = g(seed,i,k);  H[(i+d_k)%N, i] = conj(.).  Gershgorin:
    spectrum in [-rho, rho].  Returns 0-based CSR (rowptr int64, col int32,
    vals complex128) for rows [row_begin, row_end), columns ascending.  Generated in row chunks on a thread pool (NumPy
    releases the GIL inside its loops): N = 2^24 takes seconds, not minutes, and no more scratch than the chunks."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    row_end = N if row_end is None else row_end
    nk = len(offsets)
    assert 2 * max(offsets) < N, "offsets must be < N/2 so that columns are distinct"
    nloc = row_end - row_begin
    col = np.empty((nloc, 2 * nk), dtype=np.int32)
    vals = np.empty((nloc, 2 * nk), dtype=np.complex128)
    chunk = 1 << 16
    jobs = [(a, min(a + chunk, row_end)) for a in range(row_begin, row_end, chunk)]

    def run(ab):
        a, b = ab
        _offsets_rows(N, offsets, rho, seed, a, b, col[a - row_begin:b - row_begin], vals[a - row_begin:b - row_begin])
    nthreads = max(1, min(len(jobs), _usable_cpus() - 1, 16))
    if nthreads > 1:
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(run, jobs))
    else:
        for ab in jobs:
            run(ab)
    rowptr = np.arange(0, (nloc + 1) * 2 * nk, 2 * nk, dtype=np.int64)
    return rowptr, col.reshape(-1), vals.reshape(-1)


def grid_hamiltonian_2d(nx, ny, flux=0.0, next_nearest=False, seed=DEFAULT_SEED, diagonal=0.0):
    """Finite-difference Hamiltonian of a particle on an nx x ny grid with OPEN boundaries (row = x + nx y): hopping -1 to
    the four neighbours (Peierls phase exp(i flux y) on the x-hops when flux != 0, so the couplings are complex), optionally
    -1/4 to the second neighbours along x, -`diagonal` to the four diagonal neighbours, and a smooth potential on the diagonal.  The rows at the x-edges of the grid lack
    a neighbour -- the lattice with holes that operator creation completes (include/qprop.h: qp_operator_fill_info).
    Returns scipy CSR (sorted indices)."""
    import scipy.sparse as sp
    x = np.arange(nx)
    y = np.arange(ny)
    X, Y = np.meshgrid(x, y, indexing="xy")            # row-major in y: index = x + nx * y
    idx = (X + nx * Y).ravel()
    rows, cols, vals = [], [], []

    def hop(mask, d, v):
        r = idx[mask.ravel()]
        rows.extend([r, r + d])
        cols.extend([r + d, r])
        vv = v.ravel()[mask.ravel()] if np.ndim(v) else np.full(len(r), v, dtype=np.complex128)
        vals.extend([vv, np.conj(vv)])

    phase = np.exp(1j * flux * Y) if flux else np.ones_like(Y, dtype=np.complex128)
    hop(X < nx - 1, 1, -phase)
    hop(Y < ny - 1, nx, np.complex128(-1.0))
    if next_nearest:
        hop(X < nx - 2, 2, -0.25 * phase * phase)
    if diagonal:      # t' hopping to the four diagonal neighbours: distances +-(nx - 1), +-(nx + 1) -- the nine-point stencil
        hop((X < nx - 1) & (Y < ny - 1), nx + 1, -diagonal * phase)
        hop((X > 0) & (Y < ny - 1), nx - 1, -diagonal * np.conj(phase))
    with np.errstate(over="ignore"):
        jitter = _u01(splitmix64(np.uint64(seed) ^ (idx.astype(np.uint64) * _GOLDEN)))
    pot = 0.5 * ((X - nx / 2) / nx) ** 2 + 0.5 * ((Y - ny / 2) / ny) ** 2
    rows.append(idx)
    cols.append(idx)
    vals.append((4.0 + pot.ravel() + 0.01 * jitter).astype(np.complex128))
    H = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nx * ny, nx * ny))
    H.sum_duplicates()
    H.sort_indices()
    return H


def grid_hamiltonian_3d(nx, ny, nz, flux=0.0, seed=DEFAULT_SEED, order=2, diagonal=0.0):
    """Seven-point finite-difference Hamiltonian on an nx x ny x nz grid with OPEN boundaries (row = x + nx (y + ny z)):
    hopping -1 to the six neighbours (phase exp(i flux y) on the x-hops when flux != 0), a smooth potential on the diagonal.
    Distances +-1, +-nx, +-nx ny: the walk's lattice with one long pair beyond the ring (kernels_walk.hip, XL) once its edge
    rows are completed.  order = 4: the fourth-order Laplacian's thirteen points -- second neighbours along every axis with
    weight +1/12 of... (-1/12 : 4/3 stencil, scaled to the hoppings here): distances +-1, +-2, +-nx, +-2 nx, +-nx ny,
    +-2 nx ny (near 2, far 2, two long pairs).  Returns scipy CSR (sorted indices)."""
    import scipy.sparse as sp
    N = nx * ny * nz
    idx = np.arange(N, dtype=np.int64)
    X = idx % nx
    Y = (idx // nx) % ny
    Z = idx // (nx * ny)
    rows, cols, vals = [], [], []

    def hop(mask, d, v):
        r = idx[mask]
        vv = v[mask] if np.ndim(v) else np.full(len(r), v, dtype=np.complex128)
        rows.extend([r, r + d])
        cols.extend([r + d, r])
        vals.extend([vv, np.conj(vv)])

    phase = np.exp(1j * flux * Y) if flux else np.ones(N, dtype=np.complex128)
    hop(X < nx - 1, 1, -phase)
    hop(Y < ny - 1, nx, np.complex128(-1.0))
    hop(Z < nz - 1, nx * ny, np.complex128(-0.5))
    if diagonal:        # in-plane next-nearest hopping t' (layered t-t' planes coupled along z): distances +-(nx - 1), +-(nx + 1)
        hop((X < nx - 1) & (Y < ny - 1), nx + 1, -diagonal * phase)
        hop((X > 0) & (Y < ny - 1), nx - 1, -diagonal * np.conj(phase))
    if order == 4:      # -1/12 f(x +- 2 h) + 4/3 f(x +- h) - 5/2 f(x): the second neighbours carry -1/16 of the first ones' weight
        hop(X < nx - 2, 2, phase * phase / 16.0)
        hop(Y < ny - 2, 2 * nx, np.complex128(1.0 / 16.0))
        hop(Z < nz - 2, 2 * nx * ny, np.complex128(0.5 / 16.0))
    with np.errstate(over="ignore"):
        jitter = _u01(splitmix64(np.uint64(seed) ^ (idx.astype(np.uint64) * _GOLDEN)))
    pot = 0.5 * ((X - nx / 2) / nx) ** 2 + 0.5 * ((Y - ny / 2) / ny) ** 2 + 0.5 * ((Z - nz / 2) / nz) ** 2
    rows.append(idx)
    cols.append(idx)
    vals.append((5.0 + pot + 0.01 * jitter).astype(np.complex128))
    H = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))
    H.sum_duplicates()
    H.sort_indices()
    return H


def random_columns_csr(N, n_pairs=8, window=None, rho=10.0, seed=DEFAULT_SEED):
    """Hermitian H whose columns are drawn PER ROW (no translation invariance: the irregular case that
    plain "CSR" implies), 2 n_pairs entries per row, no diagonal.  Rows are cut into blocks of `window`
    consecutive rows (default: one block = all N, columns anywhere); inside a block, pairing k couples
    row i with pi_k(i), where pi_k is a single cycle through the block in a seeded random order -- so every
    row gets exactly one partner forwards and one backwards per pairing, H[i, pi_k(i)] = g,
    H[pi_k(i), i] = conj(g).  Two pairings giving a row the same partner are merged (a row then has
    fewer than 2 n_pairs entries: about n_pairs^2 / window of the rows).  |g| <= rho / (2 n_pairs):
    Gershgorin keeps the spectrum inside [-rho, rho].  Returns 0-based CSR (rowptr int64, col int32,
    vals complex128), columns ascending."""
    W = N if window is None else int(window)
    assert N % W == 0 and W > 2
    rng = np.random.default_rng(seed)
    rows_all, cols_all, vals_all = [], [], []
    nblk = N // W
    for k in range(n_pairs):
        # one random cyclic order per block: sigma[b] lists the block's rows; pi(sigma[j]) = sigma[j + 1]
        sigma = np.argsort(rng.random((nblk, W)), axis=1).astype(np.int64)
        src = (sigma + (np.arange(nblk, dtype=np.int64) * W)[:, None]).reshape(-1)
        dst = (np.roll(sigma, -1, axis=1) + (np.arange(nblk, dtype=np.int64) * W)[:, None]).reshape(-1)
        g = coupling(seed, src, k, rho, n_pairs)
        rows_all += [src, dst]
        cols_all += [dst, src]
        vals_all += [g, np.conj(g)]
    A = sp.coo_matrix((np.concatenate(vals_all), (np.concatenate(rows_all), np.concatenate(cols_all))), shape=(N, N)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.complex128)


def random_state(N, seed=DEFAULT_SEED + 1, row_begin=0, row_end=None, normalize=True):
    """Complex Gaussian (Box-Muller on the hash) state, normalised over all N."""
    def part(a, b):
        i = np.arange(a, b, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h1 = splitmix64(np.uint64(seed) ^ (i * _GOLDEN))
        h2 = splitmix64(h1)
        u1 = 1.0 - _u01(h1)
        u2 = _u01(h2)
        r = np.sqrt(-2.0 * np.log(u1))
        return r * np.exp(2j * np.pi * u2)
    row_end = N if row_end is None else row_end
    psi = part(row_begin, row_end)
    if normalize:
        nrm2 = 0.0
        step = 1 << 22
        for a in range(0, N, step):
            p = part(a, min(N, a + step))
            nrm2 += float(np.vdot(p, p).real)
        psi = psi / np.sqrt(nrm2)
    return psi.astype(np.complex128)


def to_scipy(rowptr, col, vals, n_cols):
    return sp.csr_matrix((vals, col, rowptr), shape=(len(rowptr) - 1, n_cols))


# ---- test-size dense / random-sparse fixtures (NumPy Generator; not counter based)

def dense_hermitian(N, rho=10.0, rng=None):
    """GUE-like Hermitian with spectral radius ~ rho (restates what the
    reference tests draw from QuantumControlTestUtils.random_matrix)."""
    rng = np.random.default_rng(0) if rng is None else rng
    X = (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))) / np.sqrt(2)
    H = (X + X.conj().T) / 2
    return H * (rho / (2 * np.sqrt(N) * np.sqrt(0.5)))


def dense_nonhermitian(N, rho=10.0, rng=None):
    """Ginibre matrix scaled to spectral radius ~ rho (circular law)."""
    rng = np.random.default_rng(0) if rng is None else rng
    X = (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))) / np.sqrt(2)
    return X * (rho / np.sqrt(N))


def sparse_random(N, density, rho=10.0, hermitian=False, rng=None):
    rng = np.random.default_rng(0) if rng is None else rng
    nnz_target = int(density * N * N)
    r = rng.integers(0, N, nnz_target)
    c = rng.integers(0, N, nnz_target)
    v = (rng.standard_normal(nnz_target) + 1j * rng.standard_normal(nnz_target)) / np.sqrt(2)
    A = sp.coo_matrix((v, (r, c)), shape=(N, N)).tocsr()
    A.sum_duplicates()
    if hermitian:
        A = ((A + A.conj().T) / 2).tocsr()
    scale = rho / (np.sqrt(N * density) if not hermitian else 2 * np.sqrt(N * density * 0.5))
    A = (A * scale).tocsr()
    A.sort_indices()
    return A


# ---- Liouvillian (restates src/generators.jl:473-524 formulas; host-side model build)

def ham_to_superop(H, convention="TDSE"):
    """L = 1 (x) H - H^T (x) 1  (src/generators.jl:473-486)."""
    H = sp.csr_matrix(H, dtype=np.complex128)
    Id = sp.identity(H.shape[0], dtype=np.complex128, format="csr")
    L = sp.kron(Id, H) - sp.kron(H.T, Id)
    return (L if convention == "TDSE" else 1j * L).tocsr()


def lindblad_to_superop(A, convention="TDSE"):
    """D = conj(A) (x) A - (1 (x) A^+A)/2 - ((A^+A)^T (x) 1)/2
    (src/generators.jl:493-509)."""
    A = sp.csr_matrix(A, dtype=np.complex128)
    Ad = A.conj().T
    AdT = Ad.T
    AdA = Ad @ A
    Id = sp.identity(A.shape[0], dtype=np.complex128, format="csr")
    D = sp.kron(AdT, A) - sp.kron(Id, AdA) / 2 - sp.kron(AdA.T, Id) / 2
    return (1j * D if convention == "TDSE" else D).tocsr()


def liouvillian_tridiag(n, gamma=0.05, kappa=0.02, seed=DEFAULT_SEED, convention="TDSE"):
    """Config C3: H_sys = tridiagonal hopping + flat-random diagonal
    (n x n), c_ops = {sqrt(gamma) lowering, sqrt(kappa) diag dephasing}.
    Returns a scipy CSR of dimension n^2 (non-Hermitian)."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        diag = 2.0 * _u01(splitmix64(np.uint64(seed) ^ (i * _GOLDEN))) - 1.0
    hop = np.ones(n - 1)
    H = sp.diags([hop, diag, hop], [-1, 0, 1], dtype=np.complex128, format="csr")
    lower = sp.diags([np.sqrt(gamma) * np.sqrt(np.arange(1, n))], [1], dtype=np.complex128,
                     format="csr")
    deph = sp.diags([np.sqrt(kappa) * np.arange(n) / max(n - 1, 1)], [0], dtype=np.complex128,
                    format="csr")
    L = ham_to_superop(H, convention)
    L = L + lindblad_to_superop(lower, convention) + lindblad_to_superop(deph, convention)
    L = L.tocsr()
    L.sum_duplicates()
    L.sort_indices()
    return L


def tfim_csr(n, J=1.0, h=1.0, hz=0.1):
    """Transverse-field Ising chain of n spins (open ends), N = 2^n, computational basis (bit i of the row = spin i):
    H = -J sum_i sz_i sz_{i+1} - hz sum_i sz_i - h sum_i sx_i.  Every row: the diagonal and n entries at columns
    row XOR 2^i -- the structure of every qubit-register Hamiltonian (a Pauli string with k X/Y factors couples row and
    row XOR mask): NOT translation invariant (the distance is +2^i where bit i is 0, -2^i where it is 1), but 64-row blocks
    map onto 64-row blocks for i >= 6.  Returns (rowptr int64, col int32, vals complex128), columns ascending;
    spectrum inside [-(J (n - 1) + hz n + h n), +...]."""
    N = 1 << n
    r = np.arange(N, dtype=np.int64)
    bits = ((r[:, None] >> np.arange(n, dtype=np.int64)[None, :]) & 1).astype(np.int8)
    s = 1 - 2 * bits.astype(np.float64)
    diag = -J * np.sum(s[:, :-1] * s[:, 1:], axis=1) - hz * np.sum(s, axis=1)
    del bits, s
    cols = np.empty((N, n + 1), dtype=np.int64)
    cols[:, :n] = r[:, None] ^ (np.int64(1) << np.arange(n, dtype=np.int64))[None, :]
    cols[:, n] = r
    vals = np.empty((N, n + 1), dtype=np.complex128)
    vals[:, :n] = -h
    vals[:, n] = diag
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    vals = np.take_along_axis(vals, order, axis=1)
    rowptr = np.arange(N + 1, dtype=np.int64) * (n + 1)
    return rowptr, cols.reshape(-1).astype(np.int32), vals.reshape(-1)


def tfim_pauli_terms(n, J=1.0, h=1.0, hz=0.1):
    """The strings of tfim_csr(n, J, h, hz) for lib.PauliOperator: [(amplitude, (xmask, zmask)), ...] -- one term of the lazy sum."""
    out = [(-J, (0, (1 << i) | (1 << (i + 1)))) for i in range(n - 1)]
    out += [(-hz, (0, 1 << i)) for i in range(n)]
    out += [(-h, (1 << i, 0)) for i in range(n)]
    return out


def xxz_pauli_terms(n, J=1.0, delta=0.5, hz=0.05):
    """The strings of xxz_csr(n, J, delta, hz): J/2 (XX + YY) + delta ZZ per bond, hz (i + 1) / n Z_i per site."""
    out = []
    for i in range(n - 1):
        b = (1 << i) | (1 << (i + 1))
        out += [(0.5 * J, (b, 0)), (0.5 * J, (b, b)), (delta, (0, b))]      # XX, YY (x = z = both bits), ZZ
    out += [(hz * (i + 1) / n, (0, 1 << i)) for i in range(n)]
    return out


def pauli_sum_matrix(n, strings):
    """scipy CSR matrix of sum a P from explicit Kronecker products of the 2 x 2 Pauli matrices (qubit i = bit i of the index:
    the factor of qubit n - 1 comes first in the product) -- an independent construction for the tests of the mask arithmetic."""
    import scipy.sparse as sp
    sig = {"I": sp.identity(2, dtype=np.complex128, format="csr"),
           "X": sp.csr_matrix(np.array([[0, 1], [1, 0]], dtype=np.complex128)),
           "Y": sp.csr_matrix(np.array([[0, -1j], [1j, 0]], dtype=np.complex128)),
           "Z": sp.csr_matrix(np.array([[1, 0], [0, -1]], dtype=np.complex128))}
    N = 1 << n
    H = sp.csr_matrix((N, N), dtype=np.complex128)
    for amp, (xm, zm) in strings:
        M = None
        for q in range(n - 1, -1, -1):
            xb, zb = (xm >> q) & 1, (zm >> q) & 1
            f = sig["Y" if (xb and zb) else "X" if xb else "Z" if zb else "I"]
            M = f if M is None else sp.kron(M, f, format="csr")
        H = H + complex(amp) * M
    H = H.tocsr()
    H.sort_indices()
    return H


def xxz_csr(n, J=1.0, delta=0.5, hz=0.05):
    """XXZ Heisenberg chain of n spins (open ends): H = J sum_i (sx sx + sy sy)_{i,i+1} / 2 + delta sz sz + hz sum_i (i + 1) sz_i / n.
    The exchange term couples row and row XOR (3 << i) where spins i, i + 1 differ: the number of entries of a row is its number
    of domain walls + 1 -- ragged rows, no two rows of a block alike.  Returns CSR arrays as tfim_csr."""
    import scipy.sparse as sp
    N = 1 << n
    r = np.arange(N, dtype=np.int64)
    bits = ((r[:, None] >> np.arange(n, dtype=np.int64)[None, :]) & 1).astype(np.int8)
    s = 1 - 2 * bits.astype(np.float64)
    diag = delta * np.sum(s[:, :-1] * s[:, 1:], axis=1) + hz * np.sum(s * (np.arange(n) + 1)[None, :], axis=1) / n
    rows, cols = [r], [r]
    vals = [diag.astype(np.complex128)]
    for i in range(n - 1):
        differ = bits[:, i] != bits[:, i + 1]
        rr = r[differ]
        rows.append(rr)
        cols.append(rr ^ (np.int64(3) << i))
        vals.append(np.full(len(rr), J, dtype=np.complex128))
    del bits, s
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.complex128)
