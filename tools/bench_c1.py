#!/usr/bin/env python3
"""BASELINE configs[0] (C1): N = 128 dense complex Hermitian H, Chebyshev, 200 steps with alpha = 5
(the golden fixture F2, tests/golden/make_golden.py).  The reference's own CPU-runnable case: a
parity configuration, timed here for completeness -- the whole time grid as ONE persistent launch
(qp_propagate), a launch per term from the host interface, and the NumPy restatement on the host."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L, qprop_amd.propagator as P
from oracle import qp_oracle as qo   # checker / CPU timing only

f = np.load(os.path.join(ROOT, "tests", "golden", "F2_cheby_c1_dense128.npz"))
H, psi0, dt, E_min, E_max = f["H"], f["psi0"], float(f["dt"]), float(f["E_min"]), float(f["E_max"])
tlist = dt * np.arange(201)
ctx = L.Context(0)
kw = dict(method="cheby", ctx=ctx, E_min=E_min, E_max=E_max, specrange_buffer=0.0)
res = {}
for name, small in (("one persistent launch (qp_propagate)", 8192), ("general loop, launch per term", 0)):
    L.tuning_set("small_nnz", small)
    out = P.propagate(psi0, H, tlist, **kw)
    ctx.sync()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        out = P.propagate(psi0, H, tlist, **kw)
        ctx.sync()
        best = min(best or 1e9, time.perf_counter() - t0)
    out = out.numpy() if hasattr(out, "numpy") else np.asarray(out)
    err = np.linalg.norm(out - f["checkpoints"][:, -1])
    print(f"{name:40s}: {1e3 * best:7.2f} ms for 200 steps ({1e6 * best / 200:6.1f} us/step, {int(f['n_coeffs'])} coefficients)"
          f"   |psi - F2| = {err:.1e}")
L.tuning_set("small_nnz", 8192)
wrk = qo.ChebyWrk(psi0, E_max - E_min, E_min, dt)
psi = psi0.copy()
t0 = time.perf_counter()
for _ in range(200):
    qo.cheby(psi, H, dt, wrk)
tc = time.perf_counter() - t0
print(f"{'NumPy restatement on the host (1 core)':40s}: {1e3 * tc:7.2f} ms for 200 steps ({1e6 * tc / 200:6.1f} us/step)")
