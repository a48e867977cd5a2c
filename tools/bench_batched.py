#!/usr/bin/env python3
"""BASELINE configs[4]: batched 64 states x N = 2^18 CSR H (16 nnz/row), Chebyshev.  One MI355X, or
the batch split over the GPUs of a node (strong scaling: the 64 states are divided, H is replicated,
no communication -- SURVEY 8e).  Prints one JSON line: panel prop_steps/s, state-steps/s, GB/s.

    python tools/bench_batched.py --log2n 18 --batch 64 --steps 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/bench_batched.py --batch 64
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L, qprop_amd.synth as synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=18)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--nt", type=int, default=None, help="nontemporal streams in the batched kernel (knob spmm_nt)")
    ap.add_argument("--rows", type=int, default=1, help="1 = wave-per-row kernel, lane = state (default for > 32 states); 0 = state-tiled kernel")
    ap.add_argument("--offsets", default="", help="comma-separated offsets instead of the banded lattice pattern (diagnostics)")
    ap.add_argument("--rw", type=int, default=0, help="wave-per-row kernel: 0 = matrix entries through the scalar unit (default); 1, 2, 4, 8 = one entry per lane + readlane, that many rows per wavefront")
    ap.add_argument("--strip", type=int, default=0, help="row walk of the wave-per-row kernel: strip width (0 = automatic, -1 = natural order)")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if args.batch % world:
            raise SystemExit(f"--batch {args.batch} is not divisible by {world} ranks")
    N, b_total = 1 << args.log2n, args.batch
    b = b_total // world                     # this rank's share of the states
    s0 = rank * b
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=tuple(int(o) for o in args.offsets.split(","))) if args.offsets else \
        synth.hermitian_offsets_csr(N)
    ctx = L.Context(local_rank)
    L.tuning_set("spmm_tile", args.tile)
    L.tuning_set("spmm_rows", args.rows)
    L.tuning_set("spmm_strip", args.strip)
    L.tuning_set("spmm_rw", args.rw)
    if args.nt is not None:
        L.tuning_set("spmm_nt", args.nt)
    op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
    nnz = int(rp[-1])
    states = np.stack([synth.random_state(N, seed=500 + s0 + s) for s in range(b)], axis=1)
    panel = L.State(ctx, data=states.reshape(-1))
    wrk = L.ChebyWrk(ctx, N * b, 20.0, -10.0, 1.0)
    nterms = wrk.n_coeffs - 1
    for _ in range(args.warmup):
        L.cheby_batched(panel, op, 1.0, wrk, b)
    ctx.sync()
    if dist is not None:
        dist.barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        L.cheby_batched(panel, op, 1.0, wrk, b)
    ev = ctx.timer_end()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([el, ev], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, ev = float(t[0]), float(t[1])
    kern = "spmm_rows_kernel (wave per row, lane = state)" if (args.rows >= 1 and b > 32) else "csr_spmm_kernel (state-tiled)"
    alg = 20.0 * nnz + 4.0 * (N + 1) + 80.0 * N * b          # SURVEY 8d batched model per term
    per_term = ev * 1e-3 / (args.steps * nterms)
    norms = np.linalg.norm(panel.numpy().reshape(N, b), axis=0)
    # the same states one at a time through the single-state kernel, for comparison
    single = L.State(ctx, data=states[:, 0].copy())
    w1 = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
    for _ in range(3):
        L.cheby(single, op, 1.0, w1)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(20):
        L.cheby(single, op, 1.0, w1)
    ev1 = ctx.timer_end() / 20
    if rank == 0:
      print(json.dumps({
        "metric": "batched Cheby prop_step!/s, 64 states x N=2^18 CSR (BASELINE configs[4])",
        "value": args.steps / el, "unit": "panel prop_step/s", "state_steps_per_s": b_total * args.steps / el,
        "ms_per_panel_step": 1e3 * el / args.steps, "n_gpus": world, "scaling": "strong (batch split, no communication)",
        "config": {"N": N, "batch": b_total, "states_per_gpu": b, "row_walk": dict(zip(("inner_dimension", "strip_width"), op.spmm_walk(b))),
                   "kernel": kern, "nnz_per_row": nnz / N, "matvecs_per_step": nterms},
        "roofline": {"bound": "hbm", "achieved": alg / per_term / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / per_term / 1e9 / 8000.0, "algorithmic_bytes_per_launch": alg,
                     "avg_launch_us": per_term * 1e6, "kernel": kern},
        "single_state_ms_per_step_same_N": ev1, "speedup_vs_one_state_at_a_time": b * ev1 / (1e3 * el / args.steps),
        "max_norm_drift": float(np.max(np.abs(norms - 1.0)))}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
