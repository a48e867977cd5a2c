#!/usr/bin/env python3
"""The fused Chebyshev term of ONE operator with random columns through both of its kernels -- the column-blocked mirror and
the row-block kernel (knob colblock switched on the live operator) -- for `tools/pmc_diag.sh <tag> tools/pmc_colblock.py`:

    bash tools/pmc_diag.sh cb tools/pmc_colblock.py --log2n 20
    python tools/pmc_diag_summary.py cb colblock_spmv_kernel; python tools/pmc_diag_summary.py cb rbcsr_spmv_kernel
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402
import bench_points as bp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--pattern", default="random")
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--set", nargs="*", default=[], help="knob=value before the operator is created")
args = ap.parse_args()
ctx = L.Context(0)
for kv in args.set:
    k, v = kv.split("=")
    ctx.tuning_set(k, int(v))
N = 1 << args.log2n
rp, col, vals = bp.pattern_csr(args.pattern, N)
op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
print(op.colblock_info())
wrk = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
psi = L.State(ctx, data=synth.random_state(N))
keep = ctx.tuning_get("colblock")
for knob in (keep, 0):
    ctx.tuning_set("colblock", knob)
    for _ in range(args.steps):
        L.cheby(psi, op, 1.0, wrk)
    ctx.sync()
