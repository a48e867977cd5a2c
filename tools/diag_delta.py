import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L, qprop_amd.synth as synth
ctx = L.Context(0)
for N in (1000, 4096, 8192):
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    H = synth.to_scipy(rp, col, vals, N)
    x = synth.random_state(N)
    for fmt in (L.FMT_HRB, L.FMT_RBCSR, L.FMT_CSR):
        op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)], 0, fmt)
        r_, c_, v_ = op.get_csr()
        ok = np.array_equal(c_, col) and np.array_equal(v_, vals)
        y = L.State(ctx, n=N)
        op.mul(L.State(ctx, data=x), y)
        err = np.linalg.norm(y.numpy() - H @ x)
        print(N, fmt, "roundtrip", ok, "mul err", err)
