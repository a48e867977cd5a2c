"""Worker for tests/test_gpu_sharded.py: runs ShardedCheby with the product HipBackend
under the nccl (RCCL) backend and checks it against the oracle.  Launched as a subprocess
(one per rank) with RANK / WORLD_SIZE / MASTER_* in the environment."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.sharded as sharded  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    N = 4096
    offsets = (1, 2, 3, 4, 16, 32, 48, 64)
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets)
    bounds = qo.partition_rows(rp, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    sh = sharded.ShardedCheby(ctx, rp[r0:r1 + 1] - rp[r0], col[rp[r0]:rp[r1]], vals[rp[r0]:rp[r1]], N, r0, r1,
                              20.0, -10.0, 1.0, exchange=os.environ.get("QP_EXCHANGE", "auto"))
    psi0 = synth.random_state(N)
    sh.set_state(psi0[r0:r1])
    sh.step()
    sh.step()
    sh.step(backward=True)
    torch.cuda.synchronize()
    out = sh.local_state()
    H = synth.to_scipy(rp, col, vals, N)
    wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
    ref = psi0.copy()
    qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, -1.0, wrk)
    err = float(np.linalg.norm(out - ref[r0:r1]))
    print(f"rank {rank}/{world}: err={err:.3e} format={sh.op.format} exchange={sh.exchange} M={sh.M}")
    dist.destroy_process_group()
    assert err < 1e-10
    assert sh.op.format == L.FMT_HRB


if __name__ == "__main__":
    main()
