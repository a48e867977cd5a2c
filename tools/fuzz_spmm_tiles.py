#!/usr/bin/env python3
"""Fuzz the LDS-staged tiles of the batched Chebyshev term (csrc/kernels_spmm.hip: spmm_tile_kernel; plan: engine_core.hip
operator_spmm_tiles) against the row kernel (bit for bit) and the NumPy oracle (1e-10): random lattice operators -- near distances a
random subset of 1..4 (or none), far distances random multiples m g, |m| <= 4 (gaps allowed), g any value from 64 to 700 (multiples of 4
or not), with or without diagonal, periodic wrap-around (synth.hermitian_offsets_csr) or open ends (the wrapped entries removed: rows
with fewer entries at both ends), one to three terms with coefficients, sizes that are no multiple of g or of 4 g, panels of 33 to 130
states, strip widths.  Patterns the plan must refuse (a distance between 5 and 63, a far reach beyond 4 g) are drawn too: they must
run through the row kernel with the same results.  Test infrastructure: oracle/ is the checker.

    python tools/fuzz_spmm_tiles.py [n_cases] [seed]"""
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
    bad = taken = 0
    worst = 0.0
    for case in range(ncases):
        g = int(rng.choice([64, 96, 128, 130, 200, 256, 257, 512, 700]))
        near = sorted(rng.choice(np.arange(1, 5), int(rng.integers(0, 5)), replace=False).tolist())
        far = sorted(set(int(m) * g for m in rng.choice(np.arange(1, 5), int(rng.integers(1, 5)), replace=False)))
        offsets = tuple(near) + tuple(far)
        kind = int(rng.integers(0, 8))
        if kind == 0:
            offsets = offsets + (int(rng.integers(5, 60)),)          # not a lattice the tiles take
        elif kind == 1:
            offsets = tuple(near) + (g, 5 * g)                       # far reach beyond four strip steps
        steps = int(rng.integers(12, 40))
        N = g * steps + int(rng.choice([0, 0, 1, 37, g // 2, 3 * g + 5]))
        if 2 * max(offsets) >= N or N < 4096 or N > (1 << 16):
            continue
        nterms = int(rng.integers(1, 4))
        diag = bool(rng.integers(0, 2))
        open_ends = bool(rng.integers(0, 2))
        Hs, mats = [], []
        for t in range(nterms):
            rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets, rho=4.0, seed=9000 + 17 * case + t)
            H = synth.to_scipy(rp, col, vals, N).tocoo()
            if open_ends:
                keep = np.abs(H.row - H.col) <= max(offsets)
                H = sp.coo_matrix((H.data[keep], (H.row[keep], H.col[keep])), shape=(N, N))
            H = sp.csr_matrix(H)
            if diag and t == 0:
                H = sp.csr_matrix(H + sp.diags(rng.uniform(-1, 1, N)).astype(np.complex128))
            H.sort_indices()
            Hs.append(H)
            mats.append(L.Matrix.from_scipy(ctx, H))
        ncoef = int(rng.integers(0, nterms + 1))
        Op = L.Operator(ctx, mats, ncoef)
        coeffs = rng.uniform(-1.0, 1.0, ncoef)
        if ncoef:
            Op.set_coeffs(coeffs)
        drift = nterms - ncoef
        Heff = sp.csr_matrix(sum((1.0 if i < drift else coeffs[i - drift]) * Hs[i] for i in range(nterms)))
        batch = int(rng.choice([33, 40, 64, 64, 70, 128, 130]))
        states = np.stack([synth.random_state(N, seed=100 * case + s) for s in range(batch)], axis=1)
        dts = [0.5 * (1 if rng.integers(0, 3) else -1) for _ in range(2)]
        outs, info = {}, None
        for rw in (0, -1):
            ctx.tuning_set("spmm_rw", rw)
            ctx.tuning_set("spmm_strip", int(rng.choice([0, 0, 16, 32, 100])) if rw < 0 else 0)
            if rw < 0:
                info = Op.spmm_tiles(batch)
            wrk = L.ChebyWrk(ctx, N * batch, 40.0, -20.0, 0.5)
            panel = L.State(ctx, data=states.reshape(-1))
            for dt in dts:
                L.cheby_batched(panel, Op, dt, wrk, batch)
            outs[rw] = panel.numpy().reshape(N, batch)
            panel.close()
            wrk.close()
        same = bool(np.array_equal(outs[0], outs[-1]))
        err = 0.0
        for s in (0, batch - 1):
            ow = qo.ChebyWrk(states[:, s].copy(), 40.0, -20.0, 0.5)
            ref = states[:, s].copy()
            for dt in dts:
                qo.cheby(ref, Heff, dt, ow)
            err = max(err, float(np.linalg.norm(outs[-1][:, s] - ref)))
        worst = max(worst, err)
        taken += info["taken"]
        refuse = kind in (0, 1)
        if not same or err >= 1e-10 or (refuse and info["taken"]):
            bad += 1
            print(f"CASE {case}: N={N} g={g} offsets={offsets} terms={nterms} ncoef={ncoef} diag={diag} open={open_ends} batch={batch} tiles={info} "
                  f"bit-identical={same} |tiles - oracle|={err:.3e}", flush=True)
        for h in (Op, *mats):
            h.close()
    ctx.tuning_set("spmm_rw", -1)
    ctx.tuning_set("spmm_strip", 0)
    print(f"{ncases} cases drawn (seed {seed}), {taken} took the tiles, {bad} bad, worst |tiles - oracle| = {worst:.3e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
