#!/usr/bin/env python3
"""Cost of a stream marker (event record) between dependent launches: the fused Chebyshev term of
the C2 workload launched back to back from Python, (a) nothing between the launches, (b) an event
record after every launch, (c) the same plus a second stream that waits for every event."""
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
X = [L.State(ctx, data=synth.random_state(N)), L.State(ctx, n=N)]
acc = L.State(ctx, n=N)
side = torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(64)]


def run(nlaunch, marker, waiter):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for m in range(nlaunch):
        xi = m & 1
        L.cheby_term(op, X[xi], 0, X[1 - xi], X[1 - xi], acc, acc, -0.1j, 0.0, 0.0, 1e-3)
        if marker:
            e = evs[m % 64]
            e.record()
            if waiter:
                side.wait_event(e)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / nlaunch, 1e6 * t_host / nlaunch


for rep in range(2):
    for name, mk, wt in (("back to back", False, False), ("event record after each", True, False),
                         ("event record + other stream waits", True, True)):
        per, host = run(600, mk, wt)
        print(f"{name:36s}: {per:6.2f} us per launch (host enqueue {host:5.2f})", flush=True)
