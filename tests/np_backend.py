"""NumPy stand-in for the HIP backend of qprop_amd.sharded, for CPU (gloo) tests of the
partition / exchange logic.  Test infrastructure: restates the fused-term formula of
include/qprop.h:qp_cheby_term with the oracle's arithmetic."""
import numpy as np
import scipy.sparse as sp
import torch


class _Op:
    def __init__(self, A):
        self.A = A
        self.format = 1


class NpVec(np.ndarray):
    """ndarray view with the few State methods the drivers call."""

    def copy_from(self, other):
        self[:] = other
        return self

    def scal(self, alpha):
        self *= alpha
        return self


class NumpyKrylov:
    """CPU emulation of qp_krylov + the row-partitioned Arnoldi building blocks."""

    def __init__(self, n, nvec):
        self.Q = np.zeros((nvec, n), dtype=np.complex128)
        self.G = np.zeros((nvec, nvec), dtype=np.complex128)
        self.n, self.nvec = n, nvec

    def view(self, i):
        return self.Q[i].view(NpVec)

    def multidot(self, j, reduced):
        w = self.Q[j + 1]
        for i in range(j + 1):
            reduced[i] = np.vdot(self.Q[i], w)
            reduced[j + 1 + i] = np.vdot(self.Q[i], self.Q[j])

    def project(self, j, dt, reduced, hess_col, norm_partials):
        self.G[j, :j] = np.conj(reduced[j + 1: 2 * j + 1])
        h = np.zeros(j + 1, dtype=np.complex128)
        for i in range(j + 1):
            Gi = np.conj(reduced[j + 1: j + 1 + i]) if i == j else self.G[i, :i]
            h[i] = reduced[i] - np.dot(Gi, h[:i])
        w = self.Q[j + 1]
        for i in range(j + 1):
            hd = dt * h[i]
            hess_col[i] = hd
            w += (-hd / dt) * self.Q[i]
        norm_partials[:] = 0
        norm_partials[0] = np.vdot(w, w).real

    def normalize(self, j, dt, norm_min, norm_partials, hess_norm):
        h = float(np.sqrt(norm_partials.real.sum()))
        hess_norm[0] = dt * h
        hess_norm[1] = h
        if h >= norm_min:
            self.Q[j + 1] *= 1.0 / h

    def combine(self, out, use_out, s0, first, m, coefs, norm_partials=None):
        r = (s0 * out) if use_out else np.zeros_like(out)
        for i in range(m):
            r = r + coefs[i] * self.Q[first + i]
        out[:] = r
        if norm_partials is not None:
            norm_partials[:] = 0
            norm_partials[0] = np.vdot(r, r).real


class NumpyBackend:
    def zeros(self, n):
        return torch.zeros(2 * n, dtype=torch.float64)

    def index(self, idx):
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64))

    def make_operator(self, rowptr, col, vals, nloc, ncols, fmt):
        return _Op(sp.csr_matrix((vals, col, rowptr), shape=(nloc, ncols)))

    def view(self, t, lo, hi):
        return t.numpy()[2 * lo: 2 * hi].view(np.complex128).view(NpVec)   # shares memory with the tensor

    def krylov(self, n, nvec):
        return NumpyKrylov(n, nvec)

    def mul(self, op, x, y):
        y[:] = op.A @ x

    @staticmethod
    def _accumulate(defer, acc_in, a_prev, xi, v0_old, a, t, phase):
        """Epilogue of include/qprop.h qp_cheby_term incl. qp_acc_defer; None when skipped."""
        if defer is not None and defer.skip:
            return None
        nd = 0 if defer is None else defer.n_defer
        r = acc_in if acc_in is not None else a_prev * (v0_old if nd == 1 else xi)
        if nd == 2:
            r = r + defer.a_d2 * v0_old
        if nd >= 1:
            r = r + defer.a_d1 * xi
        return phase * (r + a * t)

    def term(self, op, x, xoff, v0, vout, acc_in, acc_out, c, beta, a_prev, a, phase, defer=None):
        nloc = op.A.shape[0]
        s = op.A @ x
        xi = x[xoff: xoff + nloc]
        t = c * (s - beta * xi)
        v0_old = None if v0 is None else v0.copy()
        if v0 is not None:
            t = t + v0
        if vout is not None:
            vout[:] = t
        r = self._accumulate(defer, acc_in, a_prev, xi, v0_old, a, t, phase)
        if r is not None:
            acc_out[:] = r

    # ---- boundary / interior split (CPU emulation of qp_split / qp_cheby_term_split) ----
    def make_split(self, op, send_rows):
        A = op.A
        nloc = A.shape[0]
        nblocks = (nloc + 63) // 64
        boundary = np.zeros(nblocks, dtype=bool)
        boundary[np.asarray(send_rows, dtype=np.int64) // 64] = True
        reads_ghost = np.array([(A.indices[A.indptr[r]:A.indptr[r + 1]] >= nloc).any() for r in range(nloc)])
        boundary[np.nonzero(reads_ghost)[0] // 64] = True
        rows = np.arange(nloc)
        sp_ = type("Split", (), {})()
        sp_.brows = rows[boundary[rows // 64]]
        sp_.irows = rows[~boundary[rows // 64]]
        sp_.n_boundary, sp_.n_interior = int(boundary.sum()), int((~boundary).sum())
        sp_.send_rows = np.asarray(send_rows, dtype=np.int64)
        return sp_

    def new_stream(self):
        return None

    def stream(self, s):
        import contextlib
        return contextlib.nullcontext()

    def join(self, s):
        pass

    def current_stream(self):
        return None

    def set_stream(self, s):
        pass

    def term_split(self, op, split, side, first, x, xoff, v0, vout, acc_in, acc_out, slab, c, beta, a_prev, a, phase,
                   defer=None):
        nloc = op.A.shape[0]
        v0c = None if v0 is None else v0.copy()          # in-place v0 -> v2: keep the old values for both halves
        xc = x.copy()
        acc_c = None if acc_in is None else acc_in.copy()
        for rows, poison in ((split.brows, False), (split.irows, True)):
            xx = xc.copy()
            if poison:
                xx[nloc:] = np.nan                       # interior rows must not read a ghost slot
            s = op.A[rows] @ xx
            xi = xc[xoff + rows]
            t = c * (s - beta * xi)
            if v0c is not None:
                t = t + v0c[rows]
            if vout is not None:
                vout[rows] = t
            r = self._accumulate(defer, None if acc_c is None else acc_c[rows], a_prev, xi,
                                 None if v0c is None else v0c[rows], a, t, phase)
            if r is not None:
                acc_out[rows] = r
            if not poison and slab is not None and vout is not None:
                pos = {int(r_): i for i, r_ in enumerate(split.send_rows)}
                for r_, tv in zip(rows, t):
                    if int(r_) in pos:
                        slab[pos[int(r_)]] = tv

    def write(self, t, lo, arr):
        self.view(t, lo, lo + len(arr))[:] = arr

    def read(self, t, lo, hi):
        return self.view(t, lo, hi).copy()


class NumpyPanelBackend:
    """CPU stand-in for the b-state panel of qprop_amd.sharded.BatchSplitCheby: the oracle's cheby! state by state."""

    def make(self, rowptr, col, vals, N, b, Delta, E_min, dt):
        return _NumpyPanel(rowptr, col, vals, N, b, Delta, E_min, dt)


class _NumpyPanel:
    def __init__(self, rowptr, col, vals, N, b, Delta, E_min, dt):
        from oracle import qp_oracle as qo
        self.qo = qo
        self.H = sp.csr_matrix((vals, col, rowptr), shape=(N, N))
        self.X = np.zeros((N, b), dtype=np.complex128)
        self.args = (Delta, E_min, dt)

    def write(self, states):
        self.X[:] = states

    def read(self):
        return self.X.copy()

    def step(self, dt):
        for s_ in range(self.X.shape[1]):
            psi = self.X[:, s_].copy()
            self.qo.cheby(psi, self.H, dt, self.qo.ChebyWrk(psi, *self.args))
            self.X[:, s_] = psi

    def close(self):
        pass
