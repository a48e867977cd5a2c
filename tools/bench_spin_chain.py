#!/usr/bin/env python3
"""The fused Chebyshev term for QUBIT-REGISTER Hamiltonians -- the large-N operators a quantum-control user of the reference
actually has: a Pauli string couples row and row XOR mask, so the columns are neither a lattice (the distance is +2^i or
-2^i depending on bit i of the row) nor irregular (64-row blocks map onto 64-row blocks).  Transverse-field Ising chain
(n + 1 entries in every row) and XXZ chain (ragged rows: domain walls + 1), n = 20 ... 22 spins.

    python tools/bench_spin_chain.py [--spins 20 22] > profiles/r04/spin_chains.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402
import bench_points as bp  # noqa: E402


def measure(ctx, label, rp, col, vals, N, bound, fmt, steps, knobs):
    for k, v in knobs.items():
        ctx.tuning_set(k, v)
    nnz = int(rp[-1])
    M = L.Matrix(ctx, N, N, rp, col, vals)
    op = L.Operator(ctx, [M], 0, fmt)
    Delta, E_min = 2.2 * bound, -1.1 * bound
    dt = 20.0 / Delta                       # alpha = Delta dt / 2 = 10: 31 +- a few coefficients, as the headline
    wrk = L.ChebyWrk(ctx, N, Delta, E_min, dt)
    psi = L.State(ctx, data=synth.random_state(N))
    nterms = wrk.n_coeffs - 1
    for _ in range(2):
        L.cheby(psi, op, dt, wrk)
    regions = bp.timed_regions(ctx, lambda: L.cheby(psi, op, dt, wrk), steps, 3)
    sp = bp.spread([1e3 * r[0] / (steps * nterms) for r in regions], regions)
    real = bool(np.all(vals.imag == 0.0))         # (the kernels then stream the real copy of the values: 8 bytes each)
    by = bp.cheby_layout_bytes(op, N, N, nnz, wrk.coeffs, real_copy=real)
    lay = by["layout"]
    t = sp["median"] * 1e-6
    print(f"{label:>56s} {N:9d} {nnz / N:6.1f} {bp.FMT_NAME[op.format][:6]:>6s} {bp.cheby_kernel_name(op):>22s} {sp['median']:9.1f} {sp['min']:8.1f} {sp['max']:8.1f}"
          f" {by['csr_equivalent_per_term'] / t / 1e9:9.0f} {by['csr_equivalent_per_term'] / t / 8e12:6.3f} {by['per_term'] / t / 1e9:9.0f} {by['per_term'] / t / 8e12:6.3f}"
          f"  upper sections {op.encoding_info()['upper']}, index {lay['index_bytes'] / 1e6:.1f} MB, stored {lay['stored'] / nnz:.2f} x nnz;"
          f" walk: {op.walk_reason()[1]}; |norm - 1| {abs(psi.norm() - 1.0):.1e}; build {op.build_info()['build_ms']:.0f} ms  {knobs if knobs else ''}")
    sys.stdout.flush()
    for h in (psi, wrk, op, M):
        h.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spins", type=int, nargs="+", default=[20, 22])
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    ctx = L.Context(0)
    print("# fused Chebyshev term, real couplings (the kernels stream the real copy of the values), complex fp64 states; us per term = median of 3 regions; GB/s and fraction of 8 TB/s by the contract's CSR bytes")
    print("# ((20 z + 84) N) and by the bytes of the layout as shipped")
    print(f"{'operator':>56s} {'N':>9s} {'z':>6s} {'format':>6s} {'kernel':>22s} {'us/term':>9s} {'min':>8s} {'max':>8s} {'CSR GB/s':>9s} {'frac':>6s} {'lay GB/s':>9s} {'frac':>6s}")
    for n in args.spins:
        N = 1 << n
        for name, gen, bound in (("transverse-field Ising chain", lambda: synth.tfim_csr(n), 1.0 * (n - 1) + 0.1 * n + 1.0 * n),
                                 ("XXZ chain", lambda: synth.xxz_csr(n), 1.0 * (n - 1) + 0.5 * (n - 1) + 0.05 * n)):
            rp, col, vals = gen()
            for fmt, fname, knobs in ((L.FMT_AUTO, "auto", {}), (L.FMT_RBCSR, "rbcsr, int32 columns", {"block_map": 0}), (L.FMT_HRB, "hrb", {})):
                measure(ctx, f"{name}, {n} spins, {fname}", rp, col, vals, N, bound, fmt, args.steps, knobs)
                ctx.tuning_set("block_map", 1)
            del rp, col, vals
    ctx.close()


if __name__ == "__main__":
    main()
