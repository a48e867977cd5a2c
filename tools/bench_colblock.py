#!/usr/bin/env python3
"""Fused Chebyshev term of operators with IRREGULAR columns: the column-blocked mirror (csrc/kernels_colblock.hip) against the
row-block kernel of the same operator, per size and column pattern.

    python tools/bench_colblock.py [--log2n 19 20 21 22] [--patterns random random-window scattered] > profiles/r04/colblock.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qprop_amd.lib as L  # noqa: E402
import bench_points as bp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, nargs="+", default=[18, 19, 20, 21, 22])
    ap.add_argument("--patterns", nargs="+", default=["random", "random-window"])
    ap.add_argument("--log2w", type=int, nargs="+", default=[17])
    ap.add_argument("--waves", type=int, nargs="+", default=[24])
    ap.add_argument("--rpt", type=int, nargs="+", default=[0])
    ap.add_argument("--extra", nargs="+", default=[""], help="further knob sets, each 'knob=value,knob=value' (one measured line per set)")
    ap.add_argument("--force", action="store_true", help="build the mirror whatever the sampled gathers look like (colblock = 2)")
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    ctx = L.Context(0)
    print("# fused Chebyshev term (complex fp64, 16 entries per row), us per term = median of 3 regions; GB/s of the CSR-equivalent")
    print("# bytes (20 z + 84) N; mirror = column-blocked mirror (knob colblock), row blocks = the operator's ordinary kernel")
    print(f"{'pattern':>14s} {'N':>9s} {'kernel':>22s} {'log2w':>5s} {'waves':>5s} {'blocks':>6s} {'tile':>5s} {'us/term':>9s} {'min':>8s} {'max':>8s}"
          f" {'CSR-eq GB/s':>11s} {'frac 8 TB/s':>11s} {'build ms':>9s}  own-line share")
    for pat in args.patterns:
        for ln in args.log2n:
            ctx.tuning_set("colblock", 0)
            base = bp.measure_cheby(ctx, pattern=pat, log2n=ln, steps=args.steps, warmup=2)
            print(f"{pat:>14s} {1 << ln:9d} {base['kernel']:>22s} {'-':>5s} {'-':>5s} {'-':>6s} {'-':>5s} {base['us_per_term']:9.1f} {base['us_per_term_min']:8.1f}"
                  f" {base['us_per_term_max']:8.1f} {base['csr_equivalent_gbs']:11.0f} {base['csr_equivalent_gbs'] / 8000.0:11.3f} {base['operator_build_ms']:9.0f}")
            sys.stdout.flush()
            for lw in args.log2w:
                for wv, rpt, extra in ((w_, r_, x_) for w_ in args.waves for r_ in args.rpt for x_ in args.extra):
                    for kv in filter(None, extra.split(",")):
                        ctx.tuning_set(kv.split("=")[0], int(kv.split("=")[1]))
                    ctx.tuning_set("colblock", 2 if (ln < 19 or args.force) else 1)
                    ctx.tuning_set("cb_log2w", lw)
                    r = bp.measure_cheby(ctx, pattern=pat, log2n=ln, steps=args.steps, warmup=2)
                    ci = r["column_blocked_mirror"]
                    if not ci["valid"]:
                        print(f"{pat:>14s} {1 << ln:9d} (no mirror for log2w {lw}, waves {wv}, rpt {rpt})")
                        continue
                    print(f"{pat:>14s} {1 << ln:9d} {r['kernel']:>22s} {lw:5d} {wv:5d} {ci['column_blocks']:6d} {ci['rows_per_tile']:5d} {r['us_per_term']:9.1f}"
                          f" {r['us_per_term_min']:8.1f} {r['us_per_term_max']:8.1f} {r['csr_equivalent_gbs']:11.0f} {r['csr_equivalent_gbs'] / 8000.0:11.3f}"
                          f" {r['operator_build_ms']:9.0f}  {ci['own_line_share']:.2f}   speed-up {base['us_per_term'] / r['us_per_term']:.2f} x  {extra}   |dnorm| {r['norm_drift']:.1e}")
                    sys.stdout.flush()
    ctx.close()


if __name__ == "__main__":
    main()
