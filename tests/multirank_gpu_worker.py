"""Worker of tests/test_00_multirank_gpu.py: one rank of a 2-, 3- or 8-rank row-partitioned run.

Default ("shared" transport): ALL ranks share GPU 0.  The product HIP path runs unchanged (local numbering,
Hermitian-packed local blocks, boundary/interior split on two streams, fused pack); only the collective is staged
through the host with gloo, because RCCL cannot form a communicator of several ranks on one device.

QP_REAL_GPUS=1 ("rccl" transport): ONE GPU PER RANK -- rank r on device r, nothing staged: the library's own RCCL
communicator for the one-call step (qp_comm_prepare / qp_comm_connect), torch's nccl group for the Python-driven
schedule and the Newton reductions, gloo only for the object collectives of the set-up.  The test file picks this
mode by itself on a box with enough GPUs.  Either way the result is compared with the oracle."""
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


REAL = os.environ.get("QP_REAL_GPUS") == "1"
TRANSPORT = "rccl" if REAL else "shared"


def setup(rank, world):
    """Process group + device of this rank -> device index.  Real GPUs: device tensors travel over nccl (= RCCL), host
    objects over gloo; shared GPU: gloo only."""
    dev = rank if REAL else 0
    torch.cuda.set_device(dev)
    dist.init_process_group("cpu:gloo,cuda:nccl" if REAL else "gloo", rank=rank, world_size=world)
    return dev


def rccl_ranks(sh):
    """ncclCommCount of the communicator the native step exchanges on (0: callback stand-in or no native driver)."""
    comm = getattr(sh, "comm", None)
    return L.comm_info(comm)["rccl_ranks"] if comm is not None else 0


def newton_main(rank, world):
    dev = setup(rank, world)
    Lm = synth.liouvillian_tridiag(32)                       # N = 1024, non-Hermitian
    N = Lm.shape[0]
    bounds = qo.partition_rows(Lm.indptr.astype(np.int64), world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    ctx = L.Context(dev, stream=torch.cuda.current_stream().cuda_stream)
    lrp = Lm.indptr[r0:r1 + 1].astype(np.int64) - Lm.indptr[r0]
    sn = sharded.ShardedNewton(ctx, lrp, Lm.indices[Lm.indptr[r0]:Lm.indptr[r1]], Lm.data[Lm.indptr[r0]:Lm.indptr[r1]],
                               N, r0, r1, m_max=12, host_staged=not REAL)
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
    print(f"rank {rank}/{world}: newton err={err:.3e} restarts={sn.restarts} exchange={sn.base.exchange} transport={TRANSPORT}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not err < 1e-10:
        sys.exit(3)


def c4_main(rank, world):
    """BASELINE configs[3] at its size on ONE GPU: N = 2^24 rows, `world` ranks of 2^24 / world rows each sharing
    GPU 0, the library's one-call step (native driver) with the exchange handed back through the callback
    communicator.  Size-independent properties (norm, forward / backward round trip) and, for the state after
    ONE step, a comparison with the C restatement of the reference on windows of 2^18 rows -- after n terms a row
    depends on the rows within n x 4096 of it, so the middle of a window that wide is exact -- one window in the
    interior of a rank, one across the boundary between two ranks, one across the periodic wrap."""
    from oracle import ref_c
    dev = setup(rank, world)
    N = 1 << int(os.environ.get("QP_LOG2N", "24"))
    rows = N // world
    r0, r1 = rank * rows, (rank + 1) * rows
    rp, col, vals = synth.hermitian_offsets_csr(N, row_begin=r0, row_end=r1)
    ctx = L.Context(dev, stream=torch.cuda.current_stream().cuda_stream)
    sh = sharded.ShardedCheby(ctx, rp, col, vals, N, r0, r1, 20.0, -10.0, 1.0, exchange="auto", overlap=True,
                              host_staged=not REAL, native=True)
    del rp, col, vals
    if sh.native is None or sh.exchange != "halo" or sh.split is None:
        sys.exit(5)
    psi0 = synth.random_state(N, row_begin=r0, row_end=r1)
    sh.set_state(psi0)
    sh.step()
    torch.cuda.synchronize()
    one = sh.local_state()
    sh.step()
    torch.cuda.synchronize()
    two = sh.local_state()
    n2 = torch.tensor([float(np.vdot(one, one).real), float(np.vdot(two, two).real)], dtype=torch.float64)
    dist.all_reduce(n2)
    sh.step(backward=True)
    sh.step(backward=True)
    torch.cuda.synchronize()
    sh.check()
    back = float(np.linalg.norm(sh.local_state() - psi0))
    # windows of the oracle: (centre row, half width of the compared middle)
    W, half = 1 << 18, 2048
    coeffs = L.cheby_coeffs(20.0, 1.0)
    assert (len(coeffs) - 1) * 4096 + half < W // 2
    werr = 0.0
    for centre in (rows // 2 + 12345, 3 * rows, 0):            # interior of rank 0 | ranks 2 / 3 | ranks world-1 / 0
        lo = centre - W // 2                                  # window rows lo .. lo + W (mod N)
        wrows = (np.arange(lo, lo + W, dtype=np.int64)) % N
        mine = np.nonzero((wrows >= r0) & (wrows < r1) & (np.abs(np.arange(W) - W // 2) < half))[0]
        if len(mine) == 0:
            continue
        # the window's rows of H with columns renumbered into the window (entries leaving it dropped: they
        # only influence rows closer than 31 x 4096 to the window's edge)
        parts = []
        for a, b in ((lo % N, min(N, lo % N + W)), (0, (lo % N + W) - N)):
            if b > a:
                parts.append(synth.hermitian_offsets_csr(N, row_begin=a, row_end=b))
        wcol = np.concatenate([p_[1] for p_ in parts]).astype(np.int64)
        wval = np.concatenate([p_[2] for p_ in parts])
        wloc = (wcol - lo) % N
        keep = wloc < W
        lens = np.add.reduceat(keep.astype(np.int64), np.arange(0, len(keep), 16))
        wrp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        wpsi = np.concatenate([synth.random_state(N, row_begin=a, row_end=b) for a, b in
                               ((lo % N, min(N, lo % N + W)), (0, (lo % N + W) - N)) if b > a])
        # Hermitian window block: CSC = conj CSR
        ref_c.cheby_csc(wrp, wloc[keep], np.conj(wval[keep]), wpsi, coeffs, 20.0, -10.0, 1.0)
        werr = max(werr, float(np.max(np.abs(one[wrows[mine] - r0] - wpsi[mine]))))
    print(f"rank {rank}/{world}: c4 N=2^{int(np.log2(N))} err={werr:.3e} norm1-1={abs(float(n2[0]) - 1):.2e} "
          f"norm2-1={abs(float(n2[1]) - 1):.2e} roundtrip={back:.3e} exchange={sh.exchange} M={sh.M} p2p={sh.p2p} transport={TRANSPORT} "
          f"rccl_ranks={rccl_ranks(sh)}", flush=True)
    dist.barrier()
    sh.close()
    dist.destroy_process_group()
    if not (werr < 1e-12 and abs(float(n2[0]) - 1) < 1e-11 and abs(float(n2[1]) - 1) < 1e-11 and back < 1e-10):
        sys.exit(3)


def c4_allgather_main(rank, world):
    """BASELINE configs[3] at its size in the ALL-GATHER form -- the collective north_star names: N = 2^24 rows, `world` ranks
    sharing GPU 0, a SCATTERED Hermitian H (8 seeded offsets in [1, N/2): every row of a rank is read by another rank, so the
    send list is the whole slice and the slice itself is the send buffer), the library's one-call step with the exchange
    handed back through the callback communicator.  Checked against the C restatement of the reference at FULL size: rank 0
    regenerates the whole operator, runs one serial step (about 15 s of one core) and scatters the slices of the result."""
    from oracle import ref_c
    dev = setup(rank, world)
    N = 1 << int(os.environ.get("QP_LOG2N", "24"))
    rows = N // world
    r0, r1 = rank * rows, (rank + 1) * rows
    offs = synth.scattered_offsets(N)
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offs, row_begin=r0, row_end=r1)
    ctx = L.Context(dev, stream=torch.cuda.current_stream().cuda_stream)
    # dt = 0.1: 11 Chebyshev coefficients instead of the 32 of dt = 1 -- every term of this form moves 256 MB between the eight
    # processes through the host (the staged stand-in for the xGMI all-gather): 20 exchanges instead of 62, 1.5 instead of 5 minutes
    DT = float(os.environ.get("QP_DT", "0.1"))
    sh = sharded.ShardedCheby(ctx, rp, col, vals, N, r0, r1, 20.0, -10.0, DT, exchange="allgather", overlap=True,
                              host_staged=not REAL, native=True)
    del rp, col, vals
    if sh.native is None or sh.exchange != "allgather" or sh.M != rows:
        print(f"rank {rank}: native={sh.native is not None} exchange={sh.exchange} M={sh.M} (want the whole slice, {rows})", flush=True)
        sys.exit(5)
    psi0 = synth.random_state(N, row_begin=r0, row_end=r1)
    sh.set_state(psi0)
    sh.step()
    torch.cuda.synchronize()
    one = sh.local_state()
    n2 = torch.tensor([float(np.vdot(one, one).real)], dtype=torch.float64)
    dist.all_reduce(n2)
    sh.step(backward=True)
    torch.cuda.synchronize()
    sh.check()
    back = float(np.linalg.norm(sh.local_state() - psi0))
    mine = torch.empty(rows, 2, dtype=torch.float64)
    if rank == 0:
        coeffs = L.cheby_coeffs(20.0, DT)
        frp, fcol, fvals = synth.hermitian_offsets_csr(N, offsets=offs)
        np.conj(fvals, out=fvals)                                  # Hermitian: CSC(H) = conj CSR(H)
        ref = synth.random_state(N)
        ref_c.load()
        ref_c.cheby_csc(frp, fcol.astype(np.int64), fvals, ref, coeffs, 20.0, -10.0, DT)
        del frp, fcol, fvals
        parts = [torch.from_numpy(np.ascontiguousarray(ref[k * rows:(k + 1) * rows])).view(torch.float64).reshape(rows, 2)
                 for k in range(world)]
        dist.scatter(mine, parts, src=0)
    else:
        dist.scatter(mine, None, src=0)
    refloc = mine.numpy().view(np.complex128).reshape(-1)
    err = float(np.linalg.norm(one - refloc))
    print(f"rank {rank}/{world}: c4-allgather N=2^{int(np.log2(N))} err={err:.3e} norm1-1={abs(float(n2[0]) - 1):.2e} roundtrip={back:.3e} "
          f"exchange={sh.exchange} M={sh.M} split={'yes' if sh.split is not None else 'no'} format={sh.op.format} transport={TRANSPORT} "
          f"rccl_ranks={rccl_ranks(sh)}", flush=True)
    dist.barrier()
    sh.close()
    dist.destroy_process_group()
    if not (err < 1e-10 and abs(float(n2[0]) - 1) < 1e-11 and back < 1e-10):
        sys.exit(3)


def rccl_main(rank, world):
    """One GPU per rank, the library's own RCCL communicator (qp_comm_prepare / qp_comm_connect, id over the
    gloo group): the native one-call step in all its forms against the oracle.  Needs `world` GPUs."""
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    N = 12288
    ctx = L.Context(rank, stream=torch.cuda.current_stream().cuda_stream)
    worst = 0.0
    for offsets, exchange, overlap, p2p in (((1, 2, 3, 4, 16, 32, 48, 64), "auto", True, True),
                                            ((1, 2, 3, 4, 16, 32, 48, 64), "auto", False, False),
                                            ((5, 777, 2111, 3333, 4001, 4667, 5889, 6099), "allgather", True, False)):
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets)
        bounds = qo.partition_rows(rp, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        sh = sharded.ShardedCheby(ctx, rp[r0:r1 + 1] - rp[r0], col[rp[r0]:rp[r1]], vals[rp[r0]:rp[r1]], N, r0, r1,
                                  20.0, -10.0, 1.0, exchange=exchange, overlap=overlap, native=True, p2p=p2p)
        if sh.native is None:
            print(f"rank {rank}: native driver unavailable: {sh.native_error}", flush=True)
            sys.exit(5)
        if rccl_ranks(sh) != world:
            print(f"rank {rank}: the library's communicator has {rccl_ranks(sh)} RCCL ranks, not {world}", flush=True)
            sys.exit(6)
        psi0 = synth.random_state(N)
        sh.set_state(psi0[r0:r1])
        for _ in range(3):
            sh.step()
        sh.step(backward=True)
        torch.cuda.synchronize()
        sh.check()
        H = synth.to_scipy(rp, col, vals, N)
        wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
        ref = psi0.copy()
        for _ in range(3):
            qo.cheby(ref, H, 1.0, wrk)
        qo.cheby(ref, H, -1.0, wrk)
        worst = max(worst, float(np.linalg.norm(sh.local_state() - ref[r0:r1])))
        sh.close()
    print(f"rank {rank}/{world}: rccl err={worst:.3e} rccl_ranks={world}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not worst < 1e-10:
        sys.exit(3)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("QP_METHOD") == "newton":
        return newton_main(rank, world)
    if os.environ.get("QP_METHOD") == "c4":
        return c4_main(rank, world)
    if os.environ.get("QP_METHOD") == "c4-allgather":
        return c4_allgather_main(rank, world)
    if os.environ.get("QP_METHOD") == "rccl":
        return rccl_main(rank, world)
    overlap = os.environ.get("QP_OVERLAP", "1") == "1"
    exchange = os.environ.get("QP_EXCHANGE", "auto")
    uneven = os.environ.get("QP_UNEVEN", "0") == "1"
    native = os.environ.get("QP_NATIVE", "0") == "1"     # the library's one-call step with a callback communicator
    p2p = {"auto": "auto", "1": True, "0": False}[os.environ.get("QP_P2P", "auto")]
    dev = setup(rank, world)
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
    ctx = L.Context(dev, stream=torch.cuda.current_stream().cuda_stream)
    sh = sharded.ShardedCheby(ctx, rp[r0:r1 + 1] - rp[r0], col[rp[r0]:rp[r1]], vals[rp[r0]:rp[r1]], N, r0, r1,
                              20.0, -10.0, 1.0, exchange=exchange, overlap=overlap, host_staged=not REAL, native=native,
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
    rccl_ranks_seen = rccl_ranks(sh)
    print(f"rank {rank}/{world}: err={err:.3e} format={sh.op.format} exchange={sh.exchange} M={sh.M} "
          f"split={'yes' if sh.split is not None else 'no'} native={'yes' if sh.native is not None else 'no'} "
          f"p2p={sh.p2p} transport={TRANSPORT} rccl_ranks={rccl_ranks(sh)}", flush=True)
    dist.barrier()
    fmt_used, has_split = sh.op.format, sh.split is not None
    sh.close()      # the library's communicator goes before the process group it was bootstrapped over
    dist.destroy_process_group()
    if not err < 1e-10:
        sys.exit(3)
    if REAL and native and rccl_ranks_seen != world:
        sys.exit(6)
    if fuzz is None and exchange != "allgather" and (fmt_used != L.FMT_HRB or has_split != overlap):
        sys.exit(4)


if __name__ == "__main__":
    main()
