"""Step loop of `propagate` (src/propagate.jl:283-344) in the launch-bound regime:
  host   : prop_step! driven from Python, one launch per Chebychev term
  graph  : the same, every repeated cheby! step replayed as a hipGraph
  loop   : qp_propagate, general path (one launch per term, loop inside the library)
  loop+g : qp_propagate with the hipGraph replay
  small  : qp_propagate, one persistent single-workgroup launch for the whole time grid
Wall time per step of the loop only (init_prop excluded), state storage off."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.propagator as P  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def model(N, rng):
    if N == 512:   # 7 entries per row: register-resident
        rp, col, val = synth.hermitian_offsets_csr(N, (1, 2, 16), rho=3.0)
        return synth.to_scipy(rp, col, val, N), sp.identity(N, dtype=complex, format="csr") * 0.1
    if N <= 256:
        return synth.dense_hermitian(N, rho=3.0, rng=rng), synth.dense_hermitian(N, rho=1.0, rng=rng)
    rp, col, val = synth.hermitian_offsets_csr(N, (1, 2, 3, 16, 32, 48), rho=3.0)
    return synth.to_scipy(rp, col, val, N), sp.identity(N, dtype=complex, format="csr") * 0.1


def run(ctx, N, nt, mode, method, kw):
    rng = np.random.default_rng(0)
    H0, H1 = model(N, rng)
    tlist = np.linspace(0, 5.0, nt)
    gen = P.hamiltonian(H0, (H1, lambda t: np.cos(t)))
    psi0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi0 /= np.linalg.norm(psi0)
    L.tuning_set("small_nnz", 65536 if mode == "small" else 0)
    L.tuning_set("cheby_graph", 1024 if mode in ("graph", "loop+g", "small") else 0)
    best = None
    for _ in range(3):
        p = P.init_prop(psi0, gen, tlist, method, ctx=ctx, **kw)
        ctx.sync()
        t0 = time.perf_counter()
        if mode in ("host", "graph") or method == "newton":
            while P.prop_step(p) is not None:
                pass
        else:
            P._propagate_fused(p, False, None)
        ctx.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    L.tuning_set("small_nnz", 8192)
    L.tuning_set("cheby_graph", 0)
    return 1e6 * best / (nt - 1), p.wrk.n_coeffs if method == "cheby" else 0


def main():
    ctx = L.Context(0)
    for N, nt in ((2, 2001), (64, 2001), (128, 1001), (1024, 1001), (65536, 501)):
        for mode in ("host", "graph", "loop", "loop+g", "small"):
            us, nc = run(ctx, N, nt, mode, "cheby", dict(E_min=-20.0, E_max=20.0))
            print(f"cheby  N={N:5d} {mode:6s} {us:8.2f} us/step  ({nc} coefficients, {us / max(nc - 1, 1):6.2f} us/term)")
    for N, nt in ((64, 501), (200, 501), (512, 501)):
        for mode in ("host", "small"):
            us, _ = run(ctx, N, nt, mode, "newton", dict(m_max=10))
            print(f"newton N={N:5d} {mode:6s} {us:8.2f} us/step")
    ctx.close()


if __name__ == "__main__":
    main()
