#!/usr/bin/env python3
"""Does a second ACTIVE hardware queue slow down back-to-back dependent launches on the first one?
Plain single-stream Chebyshev steps (C2 workload), timed alone and while one idle-spinning
workgroup (torch.cuda._sleep) occupies a second stream."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qprop_amd.lib as L, qprop_amd.synth as synth

torch.cuda.set_device(0)
N = 1 << 20
rp, col, vals = synth.hermitian_offsets_csr(N)
ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
psi = L.State(ctx, data=synth.random_state(N))
wrk = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
nterms = wrk.n_coeffs - 1
side = torch.cuda.Stream(priority=int(os.environ.get("QP_PRIO", "0")))


def run(k, spin_cycles):
    for _ in range(3):
        L.cheby(psi, op, 1.0, wrk)
    torch.cuda.synchronize()
    if spin_cycles:
        with torch.cuda.stream(side):
            torch.cuda._sleep(spin_cycles)
    t0 = time.perf_counter()
    for _ in range(k):
        L.cheby(psi, op, 1.0, wrk)
    ctx.sync()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / (k * nterms)


for rep in range(2):
    print(f"alone              : {run(20, 0):.2f} us/term")
    print(f"second queue active: {run(20, 200_000_000):.2f} us/term   (one workgroup spinning on another stream)")
