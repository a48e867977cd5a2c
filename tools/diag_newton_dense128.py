import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qprop_amd.lib as L, qprop_amd.synth as synth
from oracle import qp_oracle as qo
ctx = L.Context(0)
rng = np.random.default_rng(3)
for N in (100, 128):
    H = synth.dense_hermitian(N, rho=5.0, rng=rng)
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N); psi0 /= np.linalg.norm(psi0)
    ref = psi0.copy(); ow = qo.NewtonWrk(ref, m_max=10)
    for _ in range(5): qo.newton(ref, H, 0.1, ow)
    for small in (8192, 0):
        L.tuning_set("small_nnz", small)
        Op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)])
        wrk = L.NewtonWrk(ctx, N, m_max=10)
        psi = L.State(ctx, data=psi0)
        for _ in range(5): L.newton(psi, Op, 0.1, wrk)
        err = np.linalg.norm(psi.numpy() - ref)
        for _ in range(50): L.newton(psi, Op, 0.1, wrk)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(300): L.newton(psi, Op, 0.1, wrk)
        ctx.sync(); dt = (time.perf_counter() - t0) / 300
        print(f"N={N} dense small_nnz={small}: {1e6*dt:.1f} us/step  err vs oracle after 5 steps {err:.2e}")
