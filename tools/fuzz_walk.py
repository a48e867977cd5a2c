#!/usr/bin/env python3
"""Fuzz the strip-walk kernel (csrc/kernels_walk.hip) against the NumPy oracle and the per-block kernel: random lattice
operators -- near distances up to the LDS halo, far reach 1-4 at strides of 1-40 row blocks, with or without diagonal,
real or complex couplings, one to three terms with real coefficients and a scale (`evaluate!` on a union pattern), sizes
that are no multiple of the stride or of 64, forward and backward steps, random cuts of the walk into wavefronts and
workgroup widths.  Every case that takes the walk must agree with the oracle's cheby! to 1e-10 and with the per-block
kernel bit for bit.  Test infrastructure: oracle/ is the checker.

    python tools/fuzz_walk.py [n_cases] [seed] [one case in this many carries long pairs: default 5]"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ctx = L.Context(0)
    ctx.tuning_set("walk_min_blocks", 8)
    bad = walked = 0
    worst = 0.0
    for case in range(ncases):
        nn, K = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        S = int(rng.choice([1, 2, 3, 4, 7, 16, 40]))
        g = 64 * S if rng.integers(0, 2) else int(rng.integers(64, 64 * S + 64))      # half of the cases: any stride
        near = sorted(rng.choice(np.arange(1, 17), nn, replace=False).tolist())
        nsteps = int(rng.integers(2 * K + 6, 2 * K + 60))
        N = g * nsteps + int(rng.choice([0, 0, 64, 1, 37, 200]))
        offsets = tuple(near) + tuple(g * m for m in range(1, K + 1))
        longs = int(sys.argv[3]) if len(sys.argv) > 3 else 5      # one case in `longs` carries long pairs (1: every case)
        if rng.integers(0, longs) == 0:      # the long pairs of three- and four-dimensional grids: near <= 2, far reach <= 2, one or two pairs
            nn, K = min(nn, 2), int(rng.integers(1, 3))
            near = near[:nn]
            L0 = g * (K + int(rng.integers(1, 11))) + int(rng.integers(0, 2)) * int(rng.integers(1, g))
            lo = (L0,)
            if rng.integers(0, 2):           # two pairs: the double of the first (a fourth-order stencil's second plane) or anything beyond it
                lo = (L0, 2 * L0 if rng.integers(0, 2) else L0 + int(rng.integers(1, 9 * g)))
            offsets = tuple(near) + tuple(g * m for m in range(1, K + 1)) + lo
        elif rng.integers(0, 6) == 0:        # diagonal far neighbours: g - 1, g, g + 1 (nine-point stencils), near <= 2, g >= 65
            nn, K = min(nn, 2), 1
            near = near[:nn]
            g = max(g, 65)
            N = g * nsteps + int(rng.choice([0, 0, 64, 1, 37, 200]))
            offsets = tuple(near) + (g - 1, g, g + 1)
            if rng.integers(0, 2):           # ... and one long pair beyond them (layers of such planes)
                offsets = offsets + (g * int(rng.integers(2, 12)) + int(rng.integers(18, g)),)
        if 2 * max(offsets) >= N or N > (1 << 18):
            continue
        nterms = int(rng.integers(1, 4))
        diag = bool(rng.integers(0, 2))
        real = bool(rng.integers(0, 4) == 0)
        mats, Hs = [], []
        for t in range(nterms):
            rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets, rho=6.0, seed=7000 + 13 * case + t)
            if real:
                vals = vals.real.astype(np.complex128)
            H = synth.to_scipy(rp, col, vals, N)
            if diag and t == 0:
                H = H + sp.diags(rng.uniform(-1, 1, N)).astype(np.complex128)
            H = sp.csr_matrix(H)
            H.sort_indices()
            Hs.append(H)
            mats.append(L.Matrix.from_scipy(ctx, H))
        ncoef = nterms - 1 if rng.integers(0, 2) else nterms
        ncoef = max(0, min(ncoef, nterms))
        coeffs = rng.uniform(-1.0, 1.0, ncoef)
        scale = float(rng.choice([1.0, 1.0, 0.5, -0.7]))
        Op = L.Operator(ctx, mats, ncoef, L.FMT_HRB)
        if ncoef:
            Op.set_coeffs(coeffs)
        if scale != 1.0:
            Op.set_scale(scale)
        drift = nterms - ncoef
        Heff = scale * sum((1.0 if i < drift else coeffs[i - drift]) * Hs[i] for i in range(nterms))
        Heff = sp.csr_matrix(Heff)
        wi = Op.walk_info()
        dt = float(rng.choice([0.3, 0.7, 1.1]))
        Delta, E_min = (60.0, -30.0) if rng.integers(0, 2) else (70.0, -30.0)     # (a window off centre: complex final phase)
        psi0 = synth.random_state(N, seed=case)
        wrk = L.ChebyWrk(ctx, N, Delta, E_min, dt)
        signs = [1 if rng.integers(0, 3) else -1 for _ in range(3)]
        outs = []
        for walk in (0, 1):
            ctx.tuning_set("hrb_walk", walk)
            ctx.tuning_set("walk_waves", int(rng.choice([0, 16, 64, 256, 1024, 4096])) if walk else 0)
            ctx.tuning_set("walk_pair", int(rng.choice([-1, 0, 1])) if walk else -1)      # (the two-term walk where the operator has a plan for it)
            ctx.tuning_set("walk_nt", int(rng.choice([-1, 0, 1])) if walk else -1)
            ctx.tuning_set("walk_dbg", int(rng.choice([0, 1, 4, 5])) if walk else 0)
            psi = L.State(ctx, data=psi0)
            for sg in signs:
                L.cheby(psi, Op, sg * dt, wrk)
            outs.append(psi.numpy())
        ref = psi0.copy()
        ow = qo.ChebyWrk(ref, Delta, E_min, dt)
        for sg in signs:
            qo.cheby(ref, Heff, sg * dt, ow)
        err = float(np.linalg.norm(outs[1] - ref))
        same = bool(np.array_equal(outs[0], outs[1]))
        worst = max(worst, err)
        walked += int(wi["valid"])
        if err >= 1e-10 or not same:
            bad += 1
            print(f"CASE {case}: N={N} offsets={offsets} terms={nterms} ncoef={ncoef} scale={scale} diag={diag} real={real} "
                  f"walk={wi} |walk - oracle|={err:.3e} bit-identical={same}", flush=True)
        for h in (Op, wrk, *mats):
            h.close()
    for k, v in (("hrb_walk", 1), ("walk_waves", 0), ("walk_pair", -1), ("walk_nt", -1), ("walk_dbg", 0)):
        ctx.tuning_set(k, v)
    print(f"{ncases} cases drawn (seed {seed}), {walked} took the strip walk, {bad} bad, worst |walk - oracle| = {worst:.3e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
