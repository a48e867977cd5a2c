#!/usr/bin/env python3
"""Matrix-free Liouvillian, one application L rho: the fused matrix-core kernel against the rocBLAS
zgemm chain over a range of n (where the default switches from one to the other: n = 256)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qprop_amd.lib as L, qprop_amd.synth as synth
ctx = L.Context(0)
rng = np.random.default_rng(0)
for n, nc in [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or ((192, 1), (256, 1), (320, 1), (384, 1), (512, 1), (512, 2), (640, 1), (768, 1), (1024, 1), (1024, 2)):
    H = synth.dense_hermitian(n, rho=2.0, rng=rng)
    cops = [0.2 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n) for _ in range(nc)]
    Lmf = L.Liouvillian(ctx, [H], cops, convention="TDSE")
    x = L.State(ctx, data=(rng.standard_normal(n * n) + 1j * rng.standard_normal(n * n)))
    y = L.State(ctx, n=n * n)
    res = {}
    out = {}
    for name, fused, tile in (("mfma16", 4096, 0), ("mfma32", 0, 4096), ("rocblas", 0, 0)):
        L.tuning_set("liouville_fused_n", fused)
        L.tuning_set("liouville_tile32_n", tile)
        L.tuning_set("liouville_tile32_min_n", 0)
        for _ in range(3): Lmf.mul(x, y)
        ctx.sync(); ctx.timer_begin()
        for _ in range(20): Lmf.mul(x, y)
        res[name] = 1e3 * ctx.timer_end() / 20
        out[name] = y.numpy()
    fl = 8.0 * n ** 3 * (2 + 2 * nc)
    print(f"n={n:5d} c_ops={nc}: " + "  ".join(f"{k} {v:8.1f} us ({fl / v / 1e6:5.1f} TF)" for k, v in res.items()), flush=True)
    ref = out["rocblas"]
    print("        max |diff| vs rocblas: " + "  ".join(f"{k} {np.max(np.abs(v - ref)) / np.max(np.abs(ref)):.1e}" for k, v in out.items() if k != "rocblas"), flush=True)
    Lmf.close()
L.tuning_set("liouville_fused_n", 320)
L.tuning_set("liouville_tile32_n", 2048)
L.tuning_set("liouville_tile32_min_n", 260)
