#!/usr/bin/env python3
"""Which of the two rates of the dense 64-state panel kernel (dense_zgemm_cheby_kernel, N = 4096: ~132 vs ~148 us per term, VERDICT
r04 weak 5) is the steady state?  30 timed regions of the same step in ONE process, with the shader / memory clocks read
(rocm-smi) before the first and after every fifth region, a busy-wait warm-up first and an idle pause in the middle.

    python tools/diag_dense_modes.py
"""
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L  # noqa: E402


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        m = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
        return (int(s.group(1)) if s else None, int(m.group(1)) if m else None)
    except Exception as e:  # noqa: BLE001
        return (None, str(e)[:40])


def main():
    N, batch, dt = 4096, 64, 0.5
    ctx = L.Context(0)
    rng = np.random.default_rng(7)
    X = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    H = (X + X.conj().T) * (5.0 / np.sqrt(8.0 * N))
    op = L.Operator(ctx, [L.Matrix.from_dense(ctx, H)], 0, L.FMT_DENSE)
    wrk = L.ChebyWrk(ctx, N * batch, 24.0, -12.0, dt)
    nterms = wrk.n_coeffs - 1
    st = rng.standard_normal(N * batch) + 1j * rng.standard_normal(N * batch)
    psi = L.State(ctx, data=st / np.linalg.norm(st) * np.sqrt(batch))
    print(f"dense N = {N}, {batch} states, {nterms} terms per step; clocks (sclk, mclk) MHz at start: {clocks()}")
    steps = 6
    for phase, pause in (("cold start", 0.0), ("after 3 s of back-to-back steps", 0.0), ("after a 5 s idle pause", 5.0)):
        if phase.startswith("after 3 s"):
            t0 = time.time()
            while time.time() - t0 < 3.0:
                L.cheby_batched(psi, op, dt, wrk, batch)
            ctx.sync()
        if pause:
            time.sleep(pause)
        ts = []
        for r in range(10):
            ctx.timer_begin()
            for _ in range(steps):
                L.cheby_batched(psi, op, dt, wrk, batch)
            ts.append(1e3 * ctx.timer_end() / (steps * nterms))
        print(f"{phase:34s} us per term, 10 regions of {steps} steps: " + " ".join(f"{t:6.1f}" for t in ts) + f"   clocks after: {clocks()}")
    print(f"TFLOP/s at the median of the last phase: {8.0 * N * N * batch / (np.median(ts) * 1e-6) / 1e12:.1f}")


if __name__ == "__main__":
    main()
