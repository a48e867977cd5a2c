#!/usr/bin/env python3
"""A/B timing of fused-Chebyshev-term kernel variants in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  Prints us per fused term and algorithmic GB/s.

    python tools/kbench.py --log2n 20 --variants 0,1,2,3,4,5,6,7 --rounds 7
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--n", type=int, default=0, help="explicit N (overrides --log2n)")
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--pattern", default="banded")
    ap.add_argument("--formats", default="rbcsr")
    ap.add_argument("--ab", default="", help="A/B over one knob: key=v1,v2,... (every format / variant case is run with each value)")
    ap.add_argument("--offsets", default="", help="Hermitian lattice with these distances instead of --pattern (e.g. 1,1000 or 1,2,3,4,100,200,300,400)")
    ap.add_argument("--grid", default="", help="nx,ny: finite-difference Hamiltonian on an open-boundary grid (synth.grid_hamiltonian_2d)")
    ap.add_argument("--order", type=int, default=2, help="with a three-dimensional --grid: 2 = seven-point, 4 = thirteen-point stencil")
    ap.add_argument("--no-fill", action="store_true", help="knob lattice_fill = 0 while the operator is created")
    ap.add_argument("--real", action="store_true", help="real symmetric H (values streamed as fp64 instead of complex)")
    args = ap.parse_args()
    N = args.n if args.n else 1 << args.log2n
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_points as bp
    if args.grid:
        dims = [int(t) for t in args.grid.split(",")]
        nx, ny = dims[0], dims[1]
        if len(dims) == 3:      # nx,ny,nz: seven-point grid; with --order 4 the thirteen points of the fourth-order Laplacian
            Hg = synth.grid_hamiltonian_3d(*dims, flux=0.0 if args.real else 0.1, order=args.order)
            ny = dims[1] * dims[2]
        else:
            Hg = synth.grid_hamiltonian_2d(nx, ny, flux=0.0 if args.real else 0.1)
        N = nx * ny
        rp, col, vals = Hg.indptr.astype(np.int64), Hg.indices.astype(np.int32), Hg.data.astype(np.complex128)
    elif args.offsets:
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=tuple(int(t) for t in args.offsets.split(",")))
    else:
        rp, col, vals = bp.pattern_csr(args.pattern, N)      # banded | scattered | random | random-window
    if args.real:
        vals = vals.real.astype(np.complex128)
    ctx = L.Context(0)
    if args.no_fill:
        L.tuning_set("lattice_fill", 0)
    M = L.Matrix(ctx, N, N, rp, col, vals)
    nnz = int(rp[-1])
    del rp, col, vals
    psi0 = synth.random_state(N)
    wrk = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
    nterms = wrk.n_coeffs - 1
    alg = 20.0 * nnz + 4.0 * (N + 1) + 80.0 * N
    cases = []
    for f in args.formats.split(","):
        op = L.Operator(ctx, [M], 0, {"rbcsr": L.FMT_RBCSR, "csr": L.FMT_CSR, "hrb": L.FMT_HRB}[f])
        vs = [int(v) for v in args.variants.split(",")] if f != "csr" else [0]
        for v in vs:
            cases.append((f, v, op))
    ab_key, ab_vals = None, [None]
    if args.ab:
        ab_key, vals_s = args.ab.split("=")
        ab_vals = [int(t) for t in vals_s.split(",")]
    cases = [(f, (v, av), op) for f, v, op in cases for av in ab_vals]

    def set_case(v):
        L.tuning_set("rbcsr_variant", v[0])
        if ab_key:
            L.tuning_set(ab_key, v[1])
    psi = L.State(ctx, data=psi0)
    times = {(f, v): [] for f, v, _ in cases}
    for f, v, op in cases:                      # warm-up
        set_case(v)
        L.cheby(psi, op, 1.0, wrk)
    ctx.sync()
    for r in range(args.rounds):
        for f, v, op in cases:
            set_case(v)
            ctx.timer_begin()
            for _ in range(args.steps):
                L.cheby(psi, op, 1.0, wrk)
            ms = ctx.timer_end()
            times[(f, v)].append(1e3 * ms / (args.steps * nterms))
    print(f"N={N} pattern={args.pattern} terms/step={nterms} alg_bytes/term={alg:.0f}")
    if ab_key:
        print(f"A/B knob {ab_key}: second number of 'var'")
    print(f"{'format':8s} {'var':>8s} {'median_us':>10s} {'min_us':>8s} {'csr-equiv GB/s':>15s} {'frac8T':>7s} {'layout MB':>10s} {'layout GB/s':>12s} {'frac8T':>7s}  encodings")
    ops = {(f, v): op for f, v, op in cases}
    for (f, v), t in times.items():
        med, mn = float(np.median(t)), float(np.min(t))
        by = bp.cheby_layout_bytes(ops[(f, v)], N, N, nnz, wrk.coeffs, real_copy=args.real)
        lay = by["layout"]
        vs_ = f"{v[0]}" + (f"/{v[1]}" if ab_key else "")
        print(f"{f:8s} {vs_:>8s} {med:10.2f} {mn:8.2f} {alg / med / 1e3:15.0f} {alg / med / 1e3 / 8000:7.3f} {by['per_term'] / 1e6:10.1f} "
              f"{by['per_term'] / med / 1e3:12.0f} {by['per_term'] / med / 1e3 / 8000:7.3f}  "
              f"blocks {lay['blocks']} stencil upper/lower {lay['stencil_upper_blocks']}/{lay['stencil_lower_blocks']} index bytes {lay['index_bytes']}")
    nrm = psi.norm()
    print("norm drift after all steps:", abs(nrm - 1.0))


if __name__ == "__main__":
    main()
