"""Which Newton steps of config C3 are slow, and in which host phase (qp_newton_stats)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

Lm = synth.liouvillian_tridiag(512)
N = Lm.shape[0]
ctx = L.Context(0)
op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lm)])
wrk = L.NewtonWrk(ctx, N, m_max=20)
psi = L.State(ctx, data=synth.random_state(N))
rows = []
t_all = time.perf_counter()
for k in range(150):
    t0 = time.perf_counter()
    L.newton(psi, op, 0.5, wrk)
    dt = 1e3 * (time.perf_counter() - t0)
    rows.append((dt, time.perf_counter() - t_all, {k_: round(v, 2) for k_, v in wrk.stats.items() if k_.startswith("ms_")}))
ts = np.array([r[0] for r in rows])
print("median %.2f ms, mean %.2f, max %.2f" % (np.median(ts), ts.mean(), ts.max()))
for i, r in enumerate(rows):
    if r[0] > 1.5 * np.median(ts[:20]) or i < 3:
        print(i, "t=%.3fs" % r[1], "%.2f ms" % r[0], r[2])
