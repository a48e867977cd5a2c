#!/usr/bin/env python3
"""BASELINE config C3: N = 2^18 non-Hermitian sparse Liouvillian (n = 512 system:
tridiagonal H + lowering and dephasing Lindblad operators, TDSE convention), Newton /
restarted Arnoldi with m_max = 20, on one MI355X.  Prints one JSON line with steps/s,
restarts, mat-vec count and the algorithmic-bytes model of SURVEY 8d for the restart.

    python tools/bench_newton.py --n 512 --m 20 --dt 0.5 --steps 10
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--m", type=int, default=20)
    ap.add_argument("--dt", type=float, default=0.5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=60,
                    help="untimed steps; the first ~0.2 s of a process see one-off host stalls of tens of ms on the test boxes")
    ap.add_argument("--format", default="auto", choices=["auto", "rbcsr", "csr"])
    ap.add_argument("--check", action="store_true", help="compare one step with the NumPy oracle (slow)")
    ap.add_argument("--variant", type=int, default=None, help="knob rbcsr_variant (kernel variant bits) for A/B runs")
    ap.add_argument("--pipeline", type=int, default=1, help="1 = Hessenberg eigenvalues overlap the Arnoldi sweep (default)")
    ap.add_argument("--arnoldi-mode", type=int, default=1, help="1 = low-sync MGS (default), 0 = sequential MGS passes")
    ap.add_argument("--fuse-dots", type=int, default=1, help="1 = a column's multidot runs in its mat-vec's epilogue (default), 0 = own launch")
    args = ap.parse_args()
    Lm = synth.liouvillian_tridiag(args.n)
    N = Lm.shape[0]
    nnz = Lm.nnz
    ctx = L.Context(0)
    L.tuning_set("arnoldi_mode", args.arnoldi_mode)
    L.tuning_set("arnoldi_fuse_dots", args.fuse_dots)
    if args.variant is not None:
        L.tuning_set("rbcsr_variant", args.variant)
    L.tuning_set("newton_pipeline", args.pipeline)
    fmt = {"auto": L.FMT_AUTO, "rbcsr": L.FMT_RBCSR, "csr": L.FMT_CSR}[args.format]
    op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lm)], 0, fmt)
    rho0 = synth.random_state(N)
    wrk = L.NewtonWrk(ctx, N, m_max=args.m)
    psi = L.State(ctx, data=rho0)
    parity = None
    if args.check:
        from oracle import qp_oracle as qo
        L.newton(psi, op, args.dt, wrk)
        ref = qo.newton(rho0.copy(), Lm, args.dt, qo.NewtonWrk(rho0, m_max=args.m))
        parity = float(np.linalg.norm(psi.numpy() - ref))
        psi.upload(rho0)
    for _ in range(args.warmup):
        L.newton(psi, op, args.dt, wrk)
    psi.upload(rho0)          # the open system relaxes: time the same steps every run (2-3 sweeps each at first)
    ctx.sync()
    ctx.reset_stats()
    restarts, matvecs = 0, 0
    host = {k: 0.0 for k in ("ms_arnoldi", "ms_eig", "ms_leja", "ms_coeffs", "ms_poly", "ms_update", "ms_exposed")}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        L.newton(psi, op, args.dt, wrk)
        restarts += wrk.restarts + 1          # Arnoldi sweeps = restarts + 1
        matvecs += wrk.stats["n_matvec"]
        for k in host:
            host[k] += wrk.stats[k]
    ctx.sync()
    el = time.perf_counter() - t0
    st = ctx.stats()
    m = args.m
    z = nnz / N
    # SURVEY 8d model per Arnoldi sweep (this implementation's MGS pass is 64 N, not 48 N)
    sweep_bytes = m * (20 * z + 36) * N + 64 * N * m * (m + 1) / 2 + 64 * N * m + 32 * N * m + 16 * (m + 2) * N + 16 * (m + 3) * N
    sweeps_per_s = restarts / el
    impl_bytes = m * 20 * z * N + 16 * N * m * (m + 5) + 16 * N * (m + 3)
    print(json.dumps({
        "metric": "Newton prop_step!/s, N=2^18 non-Hermitian Liouvillian, m_max=20 (BASELINE configs[2])",
        "value": args.steps / el, "unit": "prop_step/s", "ms_per_step": 1e3 * el / args.steps,
        "config": {"N": N, "nnz": nnz, "nnz_per_row": z, "m_max": m, "dt": args.dt,
                   "device_format": {1: "csr", 2: "rbcsr", 3: "hrb"}[op.format],
                   "orthogonalisation": "low-sync MGS" if args.arnoldi_mode == 1 else "sequential fused MGS passes"},
        "arnoldi_sweeps_per_step": restarts / args.steps, "matvecs_per_step": matvecs / args.steps,
        "kernel_launches_per_step": st["n_kernel_launches"] / args.steps,
        "ms_per_sweep": 1e3 / sweeps_per_s,
        # achieved / frac: the bytes this implementation moves per sweep (per column the matrix, the fused mat-vec's vectors
        # x, w, q_j and the j older basis vectors, the projection's w twice and j + 1 basis vectors; then the two combines);
        # the SURVEY 8d model counts the reference's sequential Gram-Schmidt passes and is reported beside it
        "roofline": {"bound": "hbm (Infinity-Cache resident and launch / L1-queue limited at N = 2^18)",
                     "achieved": impl_bytes * sweeps_per_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": impl_bytes * sweeps_per_s / 1e9 / 8000.0, "implementation_bytes_per_sweep": impl_bytes,
                     "survey_8d_model_gbs": sweep_bytes * sweeps_per_s / 1e9, "survey_8d_model_frac": sweep_bytes * sweeps_per_s / 1e9 / 8000.0,
                     "algorithmic_bytes_per_sweep": sweep_bytes},
        "host_ms_per_step": {k: v / args.steps for k, v in host.items()},
        "norm": psi.norm(), "parity_l2_vs_oracle_one_step": parity}))


if __name__ == "__main__":
    main()
