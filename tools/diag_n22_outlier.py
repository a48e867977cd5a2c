#!/usr/bin/env python3
"""Why did ONE of eleven round-3 bench lines show 455 us per term at N = 2^22 (122-136 in the other ten)?

Runs the banded N = 2^22 extras point of bench.py many times under the conditions that could differ between runs and
prints, per timed repeat, the HIP-event time per term NEXT TO what the host did meanwhile (wall time of the enqueue
loop, longest single `cheby!` call): a device slow mode shows as a long event time with a short enqueue; a host stall
(the enqueue loop held up while the device drains its queue and idles) shows as event time ~ enqueue time.

    python tools/diag_n22_outlier.py [--cycles 6] [--repeats 6] [--steps 5] > profiles/r04/n22_outlier.txt
"""
import argparse
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402
import bench_points as bp  # noqa: E402


def build(ctx, log2n):
    N = 1 << log2n
    rp, col, vals = bp.pattern_csr("banded", N)
    M = L.Matrix(ctx, N, N, rp, col, vals)
    del rp, col, vals
    op = L.Operator(ctx, [M], 0, L.FMT_AUTO)
    wrk = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
    psi = L.State(ctx, data=synth.random_state(N))
    return M, op, wrk, psi


def timed(ctx, psi, op, wrk, steps):
    """-> (us per term by HIP events, ms the host spent enqueueing, longest single call in ms)."""
    nterms = wrk.n_coeffs - 1
    ctx.sync()
    longest = 0.0
    thr0 = bp._cpu_throttled_ms()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        L.cheby(psi, op, 1.0, wrk)
        longest = max(longest, time.perf_counter() - t1)
    enq = time.perf_counter() - t0
    ms = ctx.timer_end()
    thr1 = bp._cpu_throttled_ms()
    if thr0 is not None and thr1 is not None and thr1 - thr0 > 1.0:
        print(f"    (CPU quota: the control group was throttled for {thr1 - thr0:.1f} ms during this region)")
    return 1e3 * ms / (steps * nterms), 1e3 * enq, 1e3 * longest


def report(tag, rows):
    us = np.array([r[0] for r in rows])
    print(f"{tag:58s} n={len(us):3d}  median {np.median(us):7.1f}  min {us.min():7.1f}  max {us.max():7.1f} us/term"
          f"   max/min {us.max() / us.min():.2f}")
    for i, (u, enq, lg) in enumerate(rows):
        if u > 1.3 * np.median(us):
            print(f"    OUTLIER repeat {i}: {u:.1f} us/term = {u * 31 * ARGS.steps / 1e3:.1f} ms of events; host enqueue loop "
                  f"{enq:.1f} ms, longest single cheby! call {lg:.1f} ms")
    sys.stdout.flush()


def main():
    global ARGS
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=6)
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--steps", type=int, default=5)
    ARGS = args = ap.parse_args()
    ctx = L.Context(0)
    allrows = []
    print(f"# banded N = 2^22, {args.steps} steps (x 31 fused terms) per timed repeat, 2 warm-up steps after every operator build")
    for c in range(args.cycles):
        # (a) as bench.py's extras sequence runs it: the 2^21 point first, closed, then a FRESH 2^22 operator
        M, op, wrk, psi = build(ctx, 21)
        for _ in range(2):
            L.cheby(psi, op, 1.0, wrk)
        r21 = [timed(ctx, psi, op, wrk, 8) for _ in range(3)]
        for h in (psi, wrk, op, M):
            h.close()
        M, op, wrk, psi = build(ctx, 22)
        for _ in range(2):
            L.cheby(psi, op, 1.0, wrk)
        rows = [timed(ctx, psi, op, wrk, args.steps) for _ in range(args.repeats)]
        report(f"cycle {c} (a) fresh 2^22 operator right after the 2^21 point", rows)
        allrows += rows
        # (b) the same operator, reused: no allocation in between
        rows = [timed(ctx, psi, op, wrk, args.steps) for _ in range(args.repeats)]
        report(f"cycle {c} (b) same operator again (no new allocation)", rows)
        allrows += rows
        # (c) cache policy of the matrix values: temporal / nontemporal (walk_nt 0 / 1; default -1 = nontemporal here)
        for nt in (0, 1):
            ctx.tuning_set("walk_nt", nt)
            rows = [timed(ctx, psi, op, wrk, args.steps) for _ in range(max(3, args.repeats // 2))]
            report(f"cycle {c} (c) walk_nt = {nt}", rows)
        ctx.tuning_set("walk_nt", -1)
        # (d) with the host's garbage collector forced in the middle of the enqueue (what a stall looks like)
        if c == 0:
            gc.collect()
            big = [np.empty(1 << 20) for _ in range(64)]
            rows = []
            for _ in range(3):
                t = timed(ctx, psi, op, wrk, args.steps)
                del big[:16]
                rows.append(t)
            report(f"cycle {c} (d) host frees 128 MB of numpy buffers between repeats", rows)
        # (e) the host side right after a LARGE operator has been destroyed (what precedes a point in bench.py's extras
        # sequence): build and close a 2^23-row operator (1 GB of device memory freed), then time the 2^22 one at once
        if c < 3:
            Mb, opb, wrkb, psib = build(ctx, 23)
            L.cheby(psib, opb, 1.0, wrkb)
            ctx.sync()
            for h in (psib, wrkb, opb, Mb):
                h.close()
            rows = [timed(ctx, psi, op, wrk, args.steps) for _ in range(args.repeats)]
            report(f"cycle {c} (e) right after a 2^23-row operator was built, used and destroyed", rows)
            worst = max(rows, key=lambda r: r[0])
            print(f"    slowest region of (e): {worst[0]:.1f} us/term, host enqueue loop {worst[1]:.1f} ms, longest single call {worst[2]:.1f} ms")
        # (f) the interpreter's cyclic garbage collector with a heap as large as a torch process's: a full collection in the
        # middle of the enqueue loop (forced here; in bench.py the allocation count of the ctypes temporaries triggers it)
        if c == 0:
            import torch  # noqa: F401  (only for the size of its heap)
            t0 = time.perf_counter()
            gc.collect()
            full_ms = 1e3 * (time.perf_counter() - t0)
            rows = []
            for k in range(4):
                nterms = wrk.n_coeffs - 1
                ctx.sync()
                ctx.timer_begin()
                t0 = time.perf_counter()
                longest = 0.0
                for s in range(args.steps):
                    t1 = time.perf_counter()
                    L.cheby(psi, op, 1.0, wrk)
                    if k % 2 == 1 and s == 1:
                        gc.collect()              # every second region: one full collection after its second step
                    longest = max(longest, time.perf_counter() - t1)
                enq = time.perf_counter() - t0
                rows.append((1e3 * ctx.timer_end() / (args.steps * nterms), 1e3 * enq, 1e3 * longest))
            report(f"cycle {c} (f) torch imported; a full gc.collect() ({full_ms:.0f} ms) inside every second region", rows)
        for h in (psi, wrk, op, M):
            h.close()
        print(f"    2^21 point of this cycle: {np.median([r[0] for r in r21]):.1f} us/term (median of 3 x 8 steps)")
    us = np.array([r[0] for r in allrows])
    bad = int(np.sum(us > 1.3 * np.median(us)))
    print(f"# all (a)+(b) repeats: {len(us)} timed regions of {args.steps} steps, median {np.median(us):.1f}, min {us.min():.1f}, "
          f"max {us.max():.1f} us/term; {bad} above 1.3 x median")


if __name__ == "__main__":
    main()
