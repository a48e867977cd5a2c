#!/usr/bin/env python3
"""Matrix-free Liouvillian: time of one application L rho for dense H and
Lindblad operators, against the fp64 matrix-core peak, and -- where it still fits -- against
the sparse n^2 x n^2 superoperator the reference would build for the same system."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

FP64_MATRIX_PEAK_TF = 78.6      # MI355X fp64 matrix = fp64 vector rate (spec)


def main():
    ctx = L.Context(0)
    rng = np.random.default_rng(0)
    sizes = ((128, 1), (256, 1), (512, 2), (1024, 2), (2048, 1))
    if len(sys.argv) > 1:      # e.g. "64,1 192,1 384,2": n,c_ops pairs; each is run with both implementations
        sizes = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
    for n, nc in sizes:
        H = synth.dense_hermitian(n, rho=2.0, rng=rng)
        cops = [0.2 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n) for _ in range(nc)]
        Lmf = L.Liouvillian(ctx, [H], cops, convention="TDSE")
        x = L.State(ctx, data=(rng.standard_normal(n * n) + 1j * rng.standard_normal(n * n)))
        y = L.State(ctx, n=n * n)
        for _ in range(3):
            Lmf.mul(x, y)
        ctx.sync()
        reps = 20
        ctx.timer_begin()
        for _ in range(reps):
            Lmf.mul(x, y)
        us = 1e3 * ctx.timer_end() / reps
        flops = 8.0 * n ** 3 * (2 + 2 * nc)
        L.tuning_set("liouville_fused_n", 0)          # the chain of rocBLAS zgemm calls, for comparison
        L.tuning_set("liouville_tile32_n", 0)
        for _ in range(3):
            Lmf.mul(x, y)
        ctx.timer_begin()
        for _ in range(reps):
            Lmf.mul(x, y)
        us_lib = 1e3 * ctx.timer_end() / reps
        L.tuning_set("liouville_fused_n", 320)
        L.tuning_set("liouville_tile32_n", 2048)
        out = {"n": n, "N": n * n, "c_ops": nc, "gemms_per_apply": 2 + 2 * nc, "us_per_apply": us,
               "us_per_apply_rocblas_chain": us_lib, "path": "mfma kernel, 32 x 32 tiles" if 260 <= n <= 2048 else "fused mfma kernel, 16 x 16 tiles" if n <= 320 else ("rocblas zgemm chain"),
               "tflops": flops / us / 1e6, "frac_fp64_matrix_peak": flops / us / 1e6 / FP64_MATRIX_PEAK_TF,
               # 1 (x) H - H^T (x) 1: 2 n^3 entries; per dense Lindblad operator n^4 + 2 n^3 more
               "sparse_superoperator_entries": 2 * n ** 3 + nc * (n ** 4 + 2 * n ** 3)}
        if n <= 128:      # the stored superoperator still fits: time it for comparison
            Lsp = synth.ham_to_superop(sp.csr_matrix(H), "TDSE")
            for A in cops:
                Lsp = Lsp + synth.lindblad_to_superop(sp.csr_matrix(A), "TDSE")
            Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lsp.tocsr())])
            for _ in range(3):
                Op.mul(x, y)
            ctx.timer_begin()
            for _ in range(reps):
                Op.mul(x, y)
            out["us_per_apply_sparse_superoperator"] = 1e3 * ctx.timer_end() / reps
            out["sparse_nnz"] = int(Op.nnz)
            Op.close()
        print(json.dumps(out), flush=True)
        Lmf.close()
    ctx.close()


if __name__ == "__main__":
    main()
