#!/usr/bin/env python3
"""A/B of the Newton (config C3) knobs in ONE process, interleaved rounds: each configuration runs the same `steps`
newton! steps from rho_0; median / min over the rounds.

    python tools/ab_newton.py [--n 512] [--rounds 7] --configs "two-pass:arnoldi_onepass=0;one-pass:arnoldi_onepass=2"
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

DEFAULT = ("two-pass:arnoldi_onepass=0;auto:arnoldi_onepass=1;one-pass:arnoldi_onepass=2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--m", type=int, default=20)
    ap.add_argument("--dt", type=float, default=0.5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--configs", default=DEFAULT)
    args = ap.parse_args()
    cfgs = []
    for part in args.configs.split(";"):
        name, kv = part.split(":")
        cfgs.append((name, [(k, int(v)) for k, v in (x.split("=") for x in kv.split(","))]))
    Lm = synth.liouvillian_tridiag(args.n)
    N = Lm.shape[0]
    ctx = L.Context(0)
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lm)])
    rho0 = synth.random_state(N)
    wrk = L.NewtonWrk(ctx, N, m_max=args.m)
    psi = L.State(ctx, data=rho0)
    for _ in range(40):
        L.newton(psi, op, args.dt, wrk)
    times = {name: [] for name, _ in cfgs}
    info = {}
    finals = {}
    for r in range(args.rounds):
        for name, kv in cfgs:
            for k, v in kv:
                ctx.tuning_set(k, v)
            psi.upload(rho0)
            for _ in range(3):                       # warm this configuration (a graph is recorded on the second sweep)
                L.newton(psi, op, args.dt, wrk)
            psi.upload(rho0)
            ctx.sync()
            ctx.reset_stats()
            sweeps = 0
            t0 = time.perf_counter()
            for _ in range(args.steps):
                L.newton(psi, op, args.dt, wrk)
                sweeps += wrk.restarts + 1
            ctx.sync()
            el = time.perf_counter() - t0
            st = ctx.stats()
            times[name].append(1e3 * el / args.steps)
            info[name] = (sweeps / args.steps, st["n_kernel_launches"] / args.steps, st.get("n_graph_launches", 0) / args.steps)
            finals[name] = psi.numpy()
    print(f"# Newton C3: N = {N} (n = {args.n}), m_max = {args.m}, dt = {args.dt}; {args.steps} steps from rho_0 per round, {args.rounds} interleaved rounds")
    print(f"{'configuration':24s} {'median ms/step':>15s} {'min':>8s} {'max':>8s} {'sweeps/step':>12s} {'launches/step':>14s} {'graph launches/step':>20s}   |psi - psi_base|")
    base = finals[cfgs[0][0]]
    for name, _ in cfgs:
        t = times[name]
        print(f"{name:24s} {np.median(t):15.3f} {min(t):8.3f} {max(t):8.3f} {info[name][0]:12.2f} {info[name][1]:14.1f} {info[name][2]:20.1f}   {np.linalg.norm(finals[name] - base):.2e}")


if __name__ == "__main__":
    main()
