"""Worker of tests/test_00_multirank_gpu.py: one rank of a 2- or 3-rank row-partitioned
Chebyshev run in which ALL ranks share GPU 0.  The product HIP path runs unchanged (local
numbering, Hermitian-packed local blocks, boundary/interior split on two streams, fused
pack); only the collective is staged through the host with gloo, because RCCL cannot form a
communicator of several ranks on one device.  Compares against the NumPy oracle."""
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


def newton_main(rank, world):
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    Lm = synth.liouvillian_tridiag(32)                       # N = 1024, non-Hermitian
    N = Lm.shape[0]
    bounds = qo.partition_rows(Lm.indptr.astype(np.int64), world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    lrp = Lm.indptr[r0:r1 + 1].astype(np.int64) - Lm.indptr[r0]
    sn = sharded.ShardedNewton(ctx, lrp, Lm.indices[Lm.indptr[r0]:Lm.indptr[r1]], Lm.data[Lm.indptr[r0]:Lm.indptr[r1]],
                               N, r0, r1, m_max=12, host_staged=True)
    rho0 = synth.random_state(N)
    sn.set_state(rho0[r0:r1])
    for dt in (0.3, 0.3, -0.3):
        sn.step(dt)
    torch.cuda.synchronize()
    ref = rho0.copy()
    owrk = qo.NewtonWrk(ref, m_max=12)
    for dt in (0.3, 0.3, -0.3):
        qo.newton(ref, Lm, dt, owrk)
    err = float(np.linalg.norm(sn.local_state() - ref[r0:r1]))
    print(f"rank {rank}/{world}: newton err={err:.3e} restarts={sn.restarts} exchange={sn.base.exchange}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not err < 1e-10:
        sys.exit(3)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("QP_METHOD") == "newton":
        return newton_main(rank, world)
    overlap = os.environ.get("QP_OVERLAP", "1") == "1"
    exchange = os.environ.get("QP_EXCHANGE", "auto")
    uneven = os.environ.get("QP_UNEVEN", "0") == "1"
    native = os.environ.get("QP_NATIVE", "0") == "1"     # the library's one-call step with a callback communicator
    p2p = {"auto": "auto", "1": True, "0": False}[os.environ.get("QP_P2P", "auto")]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    N = 12288
    offsets = (1, 2, 3, 4, 16, 32, 48, 64) if exchange != "allgather" else (5, 777, 2111, 3333, 4001, 4667, 5889, 6099)
    fuzz = os.environ.get("QP_FUZZ_SEED")
    if fuzz is not None:      # tools/fuzz_sharded.py: random size, band structure, partition and schedule
        rng = np.random.default_rng(int(fuzz))
        N = int(rng.integers(1500, 20000))
        offsets = tuple(sorted(set(int(o) for o in rng.integers(1, max(2, N // int(rng.choice([8, 64, 512]))), int(rng.integers(1, 7))))))
        overlap = bool(rng.random() < 0.6)
        native = bool(rng.random() < 0.6)
        p2p = [True, False, "auto"][int(rng.integers(0, 3))]
        exchange = ["auto", "halo", "allgather"][int(rng.integers(0, 3))]
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets)
    bounds = qo.partition_rows(rp, world).copy()
    if uneven:
        bounds[1:-1] += 100
    if fuzz is not None:
        cuts = np.sort(rng.choice(np.arange(64, N - 64), size=world - 1, replace=False))
        bounds = np.concatenate([[0], cuts, [N]]).astype(np.int64)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    sh = sharded.ShardedCheby(ctx, rp[r0:r1 + 1] - rp[r0], col[rp[r0]:rp[r1]], vals[rp[r0]:rp[r1]], N, r0, r1,
                              20.0, -10.0, 1.0, exchange=exchange, overlap=overlap, host_staged=True, native=native,
                              p2p=p2p)
    if fuzz is None and native and (sh.native is None or (p2p is True and not sh.p2p)):
        sys.exit(5)
    psi0 = synth.random_state(N)
    sh.set_state(psi0[r0:r1])
    for _ in range(3):
        sh.step()
    sh.step(backward=True)
    torch.cuda.synchronize()
    sh.check()
    out = sh.local_state()
    H = synth.to_scipy(rp, col, vals, N)
    wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
    ref = psi0.copy()
    for _ in range(3):
        qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, -1.0, wrk)
    err = float(np.linalg.norm(out - ref[r0:r1]))
    print(f"rank {rank}/{world}: err={err:.3e} format={sh.op.format} exchange={sh.exchange} M={sh.M} "
          f"split={'yes' if sh.split is not None else 'no'} native={'yes' if sh.native is not None else 'no'} "
          f"p2p={sh.p2p}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not err < 1e-10:
        sys.exit(3)
    if fuzz is None and exchange != "allgather" and (sh.op.format != L.FMT_HRB or (sh.split is not None) != overlap):
        sys.exit(4)


if __name__ == "__main__":
    main()
