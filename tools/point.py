#!/usr/bin/env python3
"""One measured point of tools/bench_points.py (what bench.py's `extras` are made of) on its own, with optional knob overrides
and an A/B over one knob in ONE process (interleaved rounds are the measurement functions' own repeats) -- replaces the per-config
scripts of rounds 2-5 (bench_newton.py, bench_batched.py, bench_dense.py, bench_liouville.py, bench_spin_chain.py).

    python tools/point.py c3                         # BASELINE configs[2]: Newton, N = 2^18 Liouvillian
    python tools/point.py c3 --n 2048 --ab arnoldi_onepass=0,2
    python tools/point.py c5 --batch 8 --set spmm_rw=1
    python tools/point.py cheby --log2n 22 --ab walk_pair=0,1
    python tools/point.py cheby --spins 20 | --grid 2048,2048 | --pattern scattered | --offsets 1,2047,2048,2049
    python tools/point.py dense --batch 64 ; python tools/point.py liouville --n 512
    python tools/point.py pauli --spins 20 [--pattern xxz]     # qubit register from its Pauli strings (no stored matrix)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qprop_amd.lib as L  # noqa: E402
import bench_points as bp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("point", choices=["c3", "c5", "cheby", "dense", "liouville", "pauli"])
    ap.add_argument("--n", type=int, default=None, help="c3 / liouville: system size n (N = n^2); dense: N")
    ap.add_argument("--log2n", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--pattern", default="banded")
    ap.add_argument("--format", default="auto")
    ap.add_argument("--grid", default="")
    ap.add_argument("--offsets", default="")
    ap.add_argument("--spins", type=int, default=None)
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--set", default="", help="knob=value,... applied to the context before the measurement")
    ap.add_argument("--ab", default="", help="knob=v1,v2,...: the point once per value")
    ap.add_argument("--keys", default="", help="comma-separated keys of the record to print (default: the timing and fraction keys)")
    args = ap.parse_args()
    ctx = L.Context(0)
    for kv in filter(None, args.set.split(",")):
        k, v = kv.split("=")
        ctx.tuning_set(k, int(v))
    kw = {k: v for k, v in (("steps", args.steps),) if v is not None}

    def measure():
        if args.point == "c3":
            return bp.measure_newton_c3(ctx, **{**kw, **({"n": args.n} if args.n else {})})
        if args.point == "c5":
            return bp.measure_batched_c5(ctx, **{**kw, **({"batch": args.batch} if args.batch else {}), **({"log2n": args.log2n} if args.log2n else {})})
        if args.point == "dense":
            return bp.measure_dense(ctx, **{**kw, **({"N": args.n} if args.n else {}), **({"batch": args.batch} if args.batch else {})})
        if args.point == "pauli":
            return bp.measure_pauli(ctx, spins=args.spins or 20, model="xxz" if args.pattern == "xxz" else "tfim", **kw)
        if args.point == "liouville":
            return bp.measure_liouville(ctx, **({"n": args.n} if args.n else {}))
        return bp.measure_cheby(ctx, pattern=args.pattern, fmt=args.format, real=args.real, spins=args.spins,
                                grid=tuple(int(t) for t in args.grid.split(",")) if args.grid else None,
                                offsets=tuple(int(t) for t in args.offsets.split(",")) if args.offsets else None,
                                **{**kw, **({"log2n": args.log2n} if args.log2n else {})})

    keys = [k for k in args.keys.split(",") if k] or ["us_per_term", "us_per_term_min", "us_per_term_max", "ms_per_step", "ms_per_step_min",
                                                       "us_per_apply", "frac", "model", "kernel", "unstable", "sweep", "tflops"]
    ab_key, ab_vals = (args.ab.split("=")[0], [int(v) for v in args.ab.split("=")[1].split(",")]) if args.ab else (None, [None])
    for v in ab_vals:
        if ab_key:
            ctx.tuning_set(ab_key, v)
        r = measure()
        print(json.dumps({**({ab_key: v} if ab_key else {}), **{k: r[k] for k in keys if k in r}}))
    ctx.close()


if __name__ == "__main__":
    main()
