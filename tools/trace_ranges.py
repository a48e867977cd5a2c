#!/usr/bin/env python3
"""One Newton step (config C3 size) and one Chebyshev step (config C2 size) with the named profiler ranges on, for
`QP_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats`: the marker table then reads like the reference's TimerOutputs
sections (test/test_timings.jl:28-30)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

ctx = L.Context(0)
ctx.tuning_set("roctx", 1)
Lm = synth.liouvillian_tridiag(512)
N = Lm.shape[0]
op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lm)])
wrk = L.NewtonWrk(ctx, N, m_max=20)
rho = L.State(ctx, data=synth.random_state(N))
for _ in range(5):
    L.newton(rho, op, 0.5, wrk)
ctx.sync()
Nc = 1 << 20
rp, col, vals = synth.hermitian_offsets_csr(Nc)
oph = L.Operator(ctx, [L.Matrix(ctx, Nc, Nc, rp, col, vals)])
cw = L.ChebyWrk(ctx, Nc, 20.0, -10.0, 1.0)
psi = L.State(ctx, data=synth.random_state(Nc))
for _ in range(5):
    L.cheby(psi, oph, 1.0, cw)
ctx.sync()
print("ok", wrk.restarts, psi.norm())
