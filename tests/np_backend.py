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

    def write(self, t, lo, arr):
        self.view(t, lo, lo + len(arr))[:] = arr

    def read(self, t, lo, hi):
        return self.view(t, lo, hi).copy()
