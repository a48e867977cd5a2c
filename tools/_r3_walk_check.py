"""Round 3: the strip-walk kernel against the per-block kernel (bit for bit) and the oracle, several lattice shapes."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qprop_amd.lib as L
import qprop_amd.synth as synth
from oracle import qp_oracle as qo

ctx = L.Context(0)
ctx.tuning_set("walk_min_blocks", 16)
ok = True
for N, offs, diag in [((1 << 15) + 192, synth.BANDED_OFFSETS, False), (1 << 16, (1, 2, 512, 1024), False),
                      (1 << 15, (1, 3, 7, 256), False), (1 << 15, (2, 128, 256, 384), False), (1 << 16, synth.BANDED_OFFSETS, True),
                      (1 << 15, (1, 2, 3, 4, 192, 384, 576, 768), False)]:
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offs)
    if diag:
        import scipy.sparse as sp
        H = synth.to_scipy(rp, col, vals, N) + sp.diags(np.linspace(-1, 1, N)).astype(np.complex128)
        H = H.tocsr(); H.sort_indices()
        rp, col, vals = H.indptr.astype(np.int64), H.indices.astype(np.int32), H.data.astype(np.complex128)
    Op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)], 0, L.FMT_HRB)
    wi = Op.walk_info()
    psi0 = synth.random_state(N)
    wrk = L.ChebyWrk(ctx, N, 24.0, -12.0, 1.0)
    outs = {}
    for w in (0, 1):
        ctx.tuning_set("hrb_walk", w)
        psi = L.State(ctx, data=psi0)
        L.cheby(psi, Op, 1.0, wrk)
        L.cheby(psi, Op, -1.0, wrk)
        L.cheby(psi, Op, 1.0, wrk)
        outs[w] = psi.numpy()
    same = np.array_equal(outs[0], outs[1])
    Hs = synth.to_scipy(rp, col, vals, N)
    ref = qo.cheby(psi0.copy(), Hs, 1.0, qo.ChebyWrk(psi0, 24.0, -12.0, 1.0))
    err = np.linalg.norm(outs[1] - ref)
    print(f"N={N} offsets={offs} diag={diag} walk={wi} bit-identical={same} max|d|={np.abs(outs[0]-outs[1]).max():.3e} |walk-oracle|={err:.3e}", flush=True)
    ok = ok and same and err < 1e-10
print("ALL OK" if ok else "FAILED")
