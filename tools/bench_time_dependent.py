"""Cost of evaluate! in a time-dependent Chebyshev step at N = 2^20 (drift + one control term, coefficient changed before
every step): a dense control term, a diagonal one (sparse: only its positions are rewritten, knob sparse_controls), the same
with the full combination forced, and the static / constant-coefficient references.

    python tools/bench_time_dependent.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import qprop_amd.lib as L
import qprop_amd.synth as synth
N = 1 << 20
ctx = L.Context(0)
rp, col, v0 = synth.hermitian_offsets_csr(N)
_, _, v1 = synth.hermitian_offsets_csr(N, seed=99)
M0 = L.Matrix(ctx, N, N, rp, col, v0)
M1 = L.Matrix(ctx, N, N, rp, col, 0.1 * v1)
Md = L.Matrix.from_scipy(ctx, sp.diags([np.linspace(-1, 1, N)], [0], format="csr", dtype=complex))
wrk = L.ChebyWrk(ctx, N, 24.0, -12.0, 1.0)
psi = L.State(ctx, data=synth.random_state(N))
for label, ops, nco, knob in (("static one term", [M0], 0, 1), ("H0 + c(t) H1 (dense control), set every step", [M0, M1], 1, 1),
                              ("H0 + c D (diagonal control), constant", [M0, Md], 1, 1),
                              ("H0 + c(t) D (diagonal control), set every step", [M0, Md], 1, 1),
                              ("H0 + c(t) D, knob sparse_controls = 0", [M0, Md], 1, 0)):
    ctx.tuning_set("sparse_controls", knob)
    op = L.Operator(ctx, ops, nco)
    if nco:
        op.set_coeffs([0.3])
    for _ in range(3):
        L.cheby(psi, op, 1.0, wrk)
    ctx.sync()
    t0 = time.perf_counter()
    steps = 40
    for k in range(steps):
        if nco and "constant" not in label:
            op.set_coeffs([0.3 + 0.001 * k])
        L.cheby(psi, op, 1.0, wrk)
    ctx.sync()
    if nco:
        ctx.sync()
        t1 = time.perf_counter()
        for k in range(20):
            op.set_coeffs([0.5 + 0.001 * k])
        ctx.sync()
        label += f" [set_coeffs alone {1e6 * (time.perf_counter() - t1) / 20:.0f} us]"
    print(f"{label:50s} {1e3 * (time.perf_counter() - t0) / steps:.3f} ms per step ({wrk.n_coeffs} coefficients, format {op.format})")
    op.close()
