"""Where a Newton step on a small (register-resident) system spends its time: host-side
phases reported by qp_newton_step (ms per step, averaged)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

ctx = L.Context(0)
N = 512
rp, col, val = synth.hermitian_offsets_csr(N, (1, 2, 16), rho=3.0)
H = synth.to_scipy(rp, col, val, N)
Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, sp.csr_matrix(H))])
rng = np.random.default_rng(0)
psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
psi0 /= np.linalg.norm(psi0)
for small in (8192, 0):
    L.tuning_set("small_nnz", small)
    wrk = L.NewtonWrk(ctx, N, m_max=10)
    psi = L.State(ctx, data=psi0)
    for _ in range(20):
        L.newton(psi, Op, 0.005, wrk)
    ctx.sync()
    acc = {}
    nst = 300
    t0 = time.perf_counter()
    for _ in range(nst):
        L.newton(psi, Op, 0.005, wrk)
        for k, v in wrk.stats.items():
            if k.startswith("ms_") or k in ("restarts", "n_matvec"):
                acc[k] = acc.get(k, 0.0) + v
    ctx.sync()
    wall = (time.perf_counter() - t0) / nst
    print(f"small_nnz={small}: wall {1e6 * wall:.1f} us/step; " +
          ", ".join(f"{k}={1e3 * v / nst:.1f}us" if k.startswith("ms_") else f"{k}={v / nst:.2f}" for k, v in acc.items()))
L.tuning_set("small_nnz", 8192)
ctx.close()
