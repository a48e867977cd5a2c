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


class NumpyBackend:
    def zeros(self, n):
        return torch.zeros(2 * n, dtype=torch.float64)

    def index(self, idx):
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64))

    def make_operator(self, rowptr, col, vals, nloc, ncols, fmt):
        return _Op(sp.csr_matrix((vals, col, rowptr), shape=(nloc, ncols)))

    def view(self, t, lo, hi):
        return t.numpy()[2 * lo: 2 * hi].view(np.complex128)   # shares memory with the tensor

    def term(self, op, x, xoff, v0, vout, acc_in, acc_out, c, beta, a_prev, a, phase):
        nloc = op.A.shape[0]
        s = op.A @ x
        xi = x[xoff: xoff + nloc]
        t = c * (s - beta * xi)
        if v0 is not None:
            t = t + v0
        if vout is not None:
            vout[:] = t
        r = (acc_in if acc_in is not None else a_prev * xi) + a * t
        acc_out[:] = phase * r

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

    def term_split(self, op, split, side, first, x, xoff, v0, vout, acc_in, acc_out, slab, c, beta, a_prev, a, phase):
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
            r = (acc_c[rows] if acc_c is not None else a_prev * xi) + a * t
            acc_out[rows] = phase * r
            if not poison and slab is not None and vout is not None:
                pos = {int(r_): i for i, r_ in enumerate(split.send_rows)}
                for r_, tv in zip(rows, t):
                    if int(r_) in pos:
                        slab[pos[int(r_)]] = tv

    def write(self, t, lo, arr):
        self.view(t, lo, lo + len(arr))[:] = arr

    def read(self, t, lo, hi):
        return self.view(t, lo, hi).copy()
