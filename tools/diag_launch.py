#!/usr/bin/env python3
"""Host-vs-device timing diagnostic for qp_cheby_step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import qprop_amd.lib as L
import qprop_amd.synth as synth

N = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
use_torch_stream = (len(sys.argv) > 2 and sys.argv[2] == "torch")
rp, col, vals = synth.hermitian_offsets_csr(N)
stream = torch.cuda.current_stream().cuda_stream if use_torch_stream else None
print("stream handle:", stream)
ctx = L.Context(0, stream=stream)
M = L.Matrix(ctx, N, N, rp, col, vals)
psi0 = synth.random_state(N)
for fmt in (L.FMT_HRB, L.FMT_RBCSR):
    op = L.Operator(ctx, [M], 0, fmt)
    wrk = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
    psi = L.State(ctx, data=psi0)
    for _ in range(3):
        L.cheby(psi, op, 1.0, wrk)
    ctx.sync()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        L.cheby(psi, op, 1.0, wrk)
    t_ret = time.perf_counter() - t0
    ctx.sync()
    t_all = time.perf_counter() - t0
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(K):
        L.cheby(psi, op, 1.0, wrk)
    ev = ctx.timer_end()
    t_ev = time.perf_counter() - t0
    # per-step synced
    t0 = time.perf_counter()
    for _ in range(K):
        L.cheby(psi, op, 1.0, wrk)
        ctx.sync()
    t_sync = time.perf_counter() - t0
    print(f"fmt={fmt}: host-return {1e3*t_ret/K:.3f} ms/step, wall(async+sync) {1e3*t_all/K:.3f} ms/step, "
          f"event {ev/K:.3f} ms/step (wall around events {1e3*t_ev/K:.3f}), per-step-synced {1e3*t_sync/K:.3f} ms/step")
