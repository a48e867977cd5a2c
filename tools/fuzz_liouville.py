#!/usr/bin/env python3
"""Random matrix-free Liouvillians (n, number of Hamiltonian terms and Lindblad operators, convention, coefficients,
scale, alpha / beta) through the three implementations of the application (16 x 16 matrix-core kernel, 32 x 32
matrix-core kernel, library chain) against the map written out in NumPy.
    python tools/fuzz_liouville.py [ncases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = L.Context(0)
rng = np.random.default_rng(seed)
bad = 0
for case in range(ncases):
    n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 47, 64, 65, 96, 100, 127, 129, 160, 191, 200, 255, 257, 300, 321]))
    nterms = int(rng.integers(0, 4))
    nc = int(rng.integers(0 if nterms else 1, 4))
    conv = str(rng.choice(["TDSE", "LvN"]))
    Hs = [synth.dense_hermitian(n, rho=float(rng.uniform(0.5, 3.0)), rng=rng) for _ in range(nterms)]
    cops = [0.3 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n) for _ in range(nc)]
    ncoeffs = int(rng.integers(0, nterms + 1)) if nterms else 0
    cvals = [complex(rng.standard_normal(), rng.standard_normal()) for _ in range(ncoeffs)]
    scale = complex(rng.standard_normal(), rng.standard_normal()) if rng.random() < 0.5 else 1.0
    alpha, beta = complex(rng.standard_normal(), rng.standard_normal()), complex(rng.standard_normal(), rng.standard_normal())
    rho = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    x = np.ascontiguousarray(rho.T).reshape(-1)
    y0 = rng.standard_normal(n * n) + 1j * rng.standard_normal(n * n)
    s_h, s_d = (1.0, 1j) if conv == "TDSE" else (1j, 1.0)
    drift = nterms - ncoeffs
    H = sum(((1.0 if l < drift else cvals[l - drift]) * Hl for l, Hl in enumerate(Hs)), np.zeros((n, n), complex))
    out = s_h * (H @ rho - rho @ H)
    for A in cops:
        G = A.conj().T @ A
        out = out + s_d * (A @ rho @ A.conj().T - 0.5 * (G @ rho + rho @ G))
    ref = scale * np.ascontiguousarray(out.T).reshape(-1)
    Lmf = L.Liouvillian(ctx, Hs, cops, ncoeffs=ncoeffs, convention=conv)
    if ncoeffs:
        Lmf.set_coeffs(cvals)
    Lmf.set_scale(scale)
    xs = L.State(ctx, data=x)
    errs = {}
    for name, fused, tile in (("mfma16", 4096, 0), ("mfma32", 0, 4096), ("library", 0, 0)):
        ctx.tuning_set("liouville_fused_n", fused)
        ctx.tuning_set("liouville_tile32_n", tile)
        ctx.tuning_set("liouville_tile32_min_n", 0)
        ys = L.State(ctx, n=n * n)
        Lmf.mul(xs, ys)
        e1 = np.linalg.norm(ys.numpy() - ref)
        ys.upload(y0)
        Lmf.mul(xs, ys, alpha, beta)
        e2 = np.linalg.norm(ys.numpy() - (beta * y0 + alpha * ref))
        errs[name] = max(e1, e2) / max(1.0, np.linalg.norm(ref) + np.linalg.norm(y0))
        ys.close()
    ok = all(e < 1e-13 for e in errs.values())
    if not ok:
        bad += 1
        print(f"BAD case {case}: n={n} nterms={nterms} ncoeffs={ncoeffs} nc={nc} {conv} errs={errs}", flush=True)
    for h in (xs, Lmf):
        h.close()
ctx.tuning_set("liouville_fused_n", 320)
ctx.tuning_set("liouville_tile32_n", 2048)
ctx.tuning_set("liouville_tile32_min_n", 260)
print(f"{ncases} cases, {bad} bad")
