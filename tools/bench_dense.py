#!/usr/bin/env python3
"""The dense operator format (QP_FMT_DENSE, csrc/kernels_dense.hip) against the same operator forced through the CSR
kernels: one state (row-sum kernel, GB/s of the 16 N^2 bytes a term must move) and a panel of states (H X on the fp64
matrix cores, TFLOP/s against the 78.6 peak).

    python tools/bench_dense.py [--sizes 1000,4096,8192] [--batch 64] > profiles/r04/dense.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qprop_amd.lib as L  # noqa: E402
import bench_points as bp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="128,1000,2048,4096,8192")
    ap.add_argument("--batch", default="8,32,64,128")
    ap.add_argument("--panel-sizes", default="1000,2048,4096")
    ap.add_argument("--no-csr", action="store_true")
    args = ap.parse_args()
    ctx = L.Context(0)
    print("# one state: fused Chebyshev term of a dense Hermitian H (complex fp64), us per term; GB/s = 16 N^2 bytes / time")
    print(f"{'N':>6s} {'format':>7s} {'us/term':>9s} {'min':>8s} {'max':>8s} {'GB/s':>8s} {'frac of 8 TB/s':>15s} {'build ms':>9s}")
    for N in (int(t) for t in args.sizes.split(",")):
        for fmt in (("dense", "csr") if not args.no_csr and N <= 4096 else ("dense",)):
            r = bp.measure_dense(ctx, N=N, fmt=fmt)
            print(f"{N:6d} {fmt:>7s} {r['us_per_term']:9.2f} {r['us_per_term_min']:8.2f} {r['us_per_term_max']:8.2f} {r['gbs']:8.0f} {r['frac']:15.3f} {r['operator_build_ms']:9.1f}")
            sys.stdout.flush()
    print("\n# panel of b states: Y = c (H X - beta X) + V0 per term; TFLOP/s = 8 N^2 b flop / time (fp64 matrix peak 78.6)")
    print(f"{'N':>6s} {'b':>4s} {'kernel':>28s} {'us/term':>9s} {'min':>8s} {'max':>8s} {'TFLOP/s':>8s} {'frac':>6s} {'GB/s (16 N^2 + 80 N b)':>23s}")
    for N in (int(t) for t in args.panel_sizes.split(",")):
        for b in (int(t) for t in args.batch.split(",")):
            for mfma in ((1, 0) if not args.no_csr and N <= 2048 else (1,)):
                r = bp.measure_dense(ctx, N=N, batch=b, mfma=mfma)
                print(f"{N:6d} {b:4d} {r['kernel']:>28s} {r['us_per_term']:9.2f} {r['us_per_term_min']:8.2f} {r['us_per_term_max']:8.2f} {r['tflops']:8.2f} {r['frac_fp64_matrix_peak']:6.3f} {r['gbs']:23.0f}")
                sys.stdout.flush()


if __name__ == "__main__":
    main()
