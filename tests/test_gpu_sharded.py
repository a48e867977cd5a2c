"""GPU test of the product multi-GPU driver (HipBackend + torch.distributed nccl = RCCL).
Only one GPU is available to the test box, so this runs world_size = 1, in-process (no
child process: a GPU-initialised parent must not exec).  It covers the torch-buffer <->
C-ABI plumbing, stream sharing, the local-numbering operator build and the RCCL
all_gather_into_tensor call shape; the multi-rank exchange logic is covered by
tests/test_sharded_gloo.py (CPU, world 2 and 3) and
tests/test_gpu_parity.py::test_cheby_term_row_partition (two shards on one GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["auto", "allgather"])
def test_sharded_hip_backend_world1(pg, exchange):
    import torch
    from oracle import qp_oracle as qo
    import qprop_amd.lib as L
    import qprop_amd.sharded as sharded
    import qprop_amd.synth as synth
    N = 4096
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange=exchange)
    assert sh.op.format == L.FMT_HRB
    psi0 = synth.random_state(N)
    sh.set_state(psi0)
    sh.step()
    sh.step()
    sh.step(backward=True)
    torch.cuda.synchronize()
    H = synth.to_scipy(rp, col, vals, N)
    wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
    ref = psi0.copy()
    qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, -1.0, wrk)
    assert np.linalg.norm(sh.local_state() - ref) < 1e-10
    # the collective the exchange uses, in the shape it uses it (views of one flat buffer)
    x = torch.zeros(2 * 48, dtype=torch.float64, device="cuda")
    x[:32] = torch.arange(32, dtype=torch.float64, device="cuda")
    slab = torch.empty(2 * 8, dtype=torch.float64, device="cuda")
    idx = torch.tensor([0, 1, 2, 3, 12, 13, 14, 15], device="cuda")
    torch.index_select(x[:32].view(-1, 2), 0, idx, out=slab.view(-1, 2))
    pg.all_gather_into_tensor(x[32:48], slab)
    torch.cuda.synchronize()
    assert x[32:48].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 24, 25, 26, 27, 28, 29, 30, 31]
    ctx.close()


@pytest.mark.parametrize("overlap,split_mode", [(True, 1), (True, 0), (True, 2), (False, 1)])
def test_sharded_overlap_machinery_world1(pg, overlap, split_mode):
    """Boundary / interior split on two HIP streams, fused pack into the slab and the RCCL
    all-gather on the side stream -- with one rank, using a forced send set (the first and
    last 200 rows), so that every piece of the multi-GPU step runs on the one GPU here."""
    import torch
    from oracle import qp_oracle as qo
    import qprop_amd.lib as L
    import qprop_amd.sharded as sharded
    import qprop_amd.synth as synth
    N = 8192
    L.tuning_set("split_mode", split_mode)
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    send = np.concatenate([np.arange(0, 200), np.arange(N - 200, N)])
    sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", overlap=overlap,
                              _debug_send_rows=send)
    assert sh.M == 400 and (sh.split is not None) == overlap
    if overlap:
        assert sh.split.n_boundary == 8 and sh.split.n_interior == N // 64 - 8
    psi0 = synth.random_state(N)
    sh.set_state(psi0)
    for _ in range(3):
        sh.step()
    sh.step(backward=True)
    torch.cuda.synchronize()
    H = synth.to_scipy(rp, col, vals, N)
    wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
    ref = psi0.copy()
    for _ in range(3):
        qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, -1.0, wrk)
    assert np.linalg.norm(sh.local_state() - ref) < 1e-10
    assert sh.n_exchanges == 4 * 31
    # the last exchanged vector is v_30 (term 30 of 31 wrote X[0] and packed its send rows):
    # the ghost slots of X[0] hold the all-gathered slab
    ghost = sh.be.read(sh.X[0], N, N + 400)
    slab = sh.be.read(sh.slab, 0, sh.M)
    assert np.array_equal(ghost, slab) and np.linalg.norm(slab) > 0
    # the fused pack (overlap) and the index_select pack (no overlap) must agree bit for bit
    other = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo",
                                 overlap=not overlap, _debug_send_rows=send)
    other.set_state(psi0)
    for _ in range(3):
        other.step()
    other.step(backward=True)
    torch.cuda.synchronize()
    assert np.array_equal(other.be.read(other.X[0], N, N + 400), ghost)
    assert np.array_equal(other.local_state(), sh.local_state())
    # an explicit (unfused) exchange packs exactly the send rows
    sh._exchange(1)
    torch.cuda.synchronize()
    x1 = sh.be.read(sh.X[1], 0, sh.ncols_local)
    assert np.array_equal(x1[N:N + 400], x1[send])
    # determinism of the two-stream schedule
    sh2 = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", overlap=overlap,
                               _debug_send_rows=send)
    sh2.set_state(psi0)
    for _ in range(3):
        sh2.step()
    sh2.step(backward=True)
    torch.cuda.synchronize()
    assert np.array_equal(sh2.local_state(), sh.local_state())
    if overlap:
        sh.split.check()          # no in-launch wait ever timed out
        sh2.split.check()
    L.tuning_set("split_mode", 2)
    ctx.close()


@pytest.mark.parametrize("split_mode,offsets", [(2, (1, 2, 3, 4, 256, 512, 768, 1024)), (1, (1, 2, 3, 4, 256, 512, 768, 1024)),
                                                (0, (1, 2, 3, 4, 256, 512, 768, 1024)), (2, (1, 2, 3, 4, 250, 500, 750, 1000)),
                                                (2, (1, 1000)), (2, "grid"), (2, (1, 2, 300, 600, 5000, 10000)), (2, (1, 999, 1000, 1001))],
                         ids=["mode2", "mode1", "mode0", "g250", "five_point_g1000", "open_grid_128x1024", "two_long_pairs", "diagonal_far_neighbours"])
def test_split_interior_as_strip_walk_world1(pg, split_mode, offsets):
    """A lattice operator's interior launch takes the strip walk (kernels_walk.hip) over the interior blocks from which no
    walked block reaches a boundary row; the rest of the interior are its edge blocks, the only ones that wait for the
    boundary launch.  One rank with a forced send set (first and last 4096 rows, like config C4's halo): every hand-off
    mode gives the bits of the serial schedule -- which walks the whole operator -- and of the per-block kernels; also for
    strip steps that are no multiple of the 64-row block, and for the round-4 shapes (two long pairs; diagonal far neighbours)."""
    import torch
    from oracle import qp_oracle as qo
    import qprop_amd.lib as L
    import qprop_amd.sharded as sharded
    import qprop_amd.synth as synth
    N = 1 << 17
    if offsets == "grid":    # open-boundary grid: the local operator's edge rows are completed at creation (lattice_fill)
        Hg = synth.grid_hamiltonian_2d(128, 1024, flux=0.15)
        rp, col, vals = Hg.indptr.astype(np.int64), Hg.indices.astype(np.int32), Hg.data.astype(np.complex128)
        offsets = (1, 128, 0)
    else:
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets)
    if len(offsets) == 2:    # hopping + on-site term
        Hd = synth.to_scipy(rp, col, vals, N) + sp.diags(np.linspace(-1.0, 1.0, N)).astype(np.complex128)
        Hd = sp.csr_matrix(Hd)
        Hd.sort_indices()
        rp, col, vals = Hd.indptr.astype(np.int64), Hd.indices.astype(np.int32), Hd.data.astype(np.complex128)
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    send = np.concatenate([np.arange(0, 4096), np.arange(N - 4096, N)])
    psi0 = synth.random_state(N)
    outs = {}
    try:
        ctx.tuning_set("walk_min_blocks", 64)
        ctx.tuning_set("split_mode", split_mode)
        for name, overlap, walk in (("overlap+walk", True, 1), ("serial+walk", False, 1), ("overlap, per-block", True, 0)):
            ctx.tuning_set("hrb_walk", walk)
            sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", overlap=overlap,
                                      _debug_send_rows=send, fmt=L.FMT_HRB)
            if overlap:
                wi = sh.split.walk_info()
                nb = N // 64
                reach = -(-max(offsets) // 64) + 1      # blocks between the boundary rows and the first walked block
                assert wi["valid"] == 1 and 64 + reach <= wi["first_block"] < wi["end_block"] <= nb - 64 - reach
                assert wi["edge_blocks"] == sh.split.n_interior - (wi["end_block"] - wi["first_block"])
            sh.set_state(psi0)
            for _ in range(3):
                sh.step()
            sh.step(backward=True)
            torch.cuda.synchronize()
            sh.check() if overlap else None
            outs[name] = sh.local_state()
            sh.close()
    finally:
        ctx.tuning_set("hrb_walk", 1)
        ctx.tuning_set("walk_min_blocks", 3072)
        ctx.tuning_set("split_mode", 2)
    assert np.array_equal(outs["overlap+walk"], outs["serial+walk"])
    assert np.array_equal(outs["overlap+walk"], outs["overlap, per-block"])
    H = synth.to_scipy(rp, col, vals, N)
    wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
    ref = psi0.copy()
    for _ in range(3):
        qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, -1.0, wrk)
    assert np.linalg.norm(outs["overlap+walk"] - ref) < 1e-10
    ctx.close()


def test_release_build_refuses_the_settings_that_change_results(pg):
    """The knobs that skip work (walk_dbg bit 1: the strip walk's edge blocks) or force a failure (split_dbg: no completion
    signal) exist in the developer flavour of the library only (csrc: make dev): the shipped library refuses them through
    every setter, and says so.  (The forced time-out itself: tests/test_00_multirank_gpu.py::test_split_wait_timeout_...)"""
    import torch
    import qprop_amd.lib as L
    assert L.load().qp_developer_build() == 0
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    for key, value in (("walk_dbg", 2), ("walk_dbg", 7), ("split_dbg", 1)):
        with pytest.raises(L.QPError, match="developer-build setting"):
            ctx.tuning_set(key, value)
        with pytest.raises(L.QPError, match="developer-build setting"):
            L.tuning_set(key, value)
    for key, value in (("walk_dbg", 1), ("walk_dbg", 4), ("walk_dbg", 5), ("split_dbg", 0)):   # bit-identical variants stay
        ctx.tuning_set(key, value)
    ctx.tuning_set("walk_dbg", 0)
    ctx.close()


def test_sharded_newton_world1(pg):
    """ShardedNewton with the product HipBackend (device Arnoldi building blocks + all-reduce
    call shape) at world 1 against the oracle and against the single-GPU qp_newton_step."""
    import torch
    from oracle import qp_oracle as qo
    import qprop_amd.lib as L
    import qprop_amd.sharded as sharded
    import qprop_amd.synth as synth
    Lm = synth.liouvillian_tridiag(24)
    N = Lm.shape[0]
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    sn = sharded.ShardedNewton(ctx, Lm.indptr.astype(np.int64), Lm.indices, Lm.data, N, 0, N, m_max=12)
    rho0 = synth.random_state(N)
    sn.set_state(rho0)
    for dt in (0.4, 0.4, -0.4):
        sn.step(dt)
    torch.cuda.synchronize()
    ref = rho0.copy()
    owrk = qo.NewtonWrk(ref, m_max=12)
    for dt in (0.4, 0.4, -0.4):
        qo.newton(ref, Lm, dt, owrk)
    assert np.linalg.norm(sn.local_state() - ref) < 1e-10
    Op = L.Operator(ctx, [L.Matrix.from_scipy(ctx, Lm)])
    wrk = L.NewtonWrk(ctx, N, m_max=12)
    rho = L.State(ctx, data=rho0)
    for dt in (0.4, 0.4, -0.4):
        L.newton(rho, Op, dt, wrk)
    assert np.linalg.norm(sn.local_state() - rho.numpy()) < 1e-12
    ctx.close()


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("exchange,p2p", [("halo", True), ("halo", False), ("allgather", False)])
def test_native_rccl_step_world1(pg, overlap, exchange, p2p):
    """qp_sharded_cheby_step: the whole partitioned cheby! as ONE library call, exchange by
    ncclAllGather on a communicator the library owns (qp_comm, RCCL resolved from torch's
    librccl.so).  World 1 here (one GPU per rank is an RCCL requirement), with a forced send
    set so that pack, all-gather, side stream and in-launch hand-off all run; the result is
    bit-identical to the torch.distributed-driven step of the same object and matches the
    oracle."""
    import torch
    from oracle import qp_oracle as qo
    import qprop_amd.lib as L
    import qprop_amd.sharded as sharded
    import qprop_amd.synth as synth
    N = 8192
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    send = np.concatenate([np.arange(0, 200), np.arange(N - 200, N)]) if exchange == "halo" else None
    sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange=exchange, overlap=overlap,
                              native=True, p2p=p2p, _debug_send_rows=send)
    assert sh.native is not None and sh.p2p == p2p      # p2p: ncclSend/ncclRecv to itself at world 1
    if exchange == "halo":
        assert sh.M == 400 and (sh.split is not None) == overlap
    psi0 = synth.random_state(N)
    results = []
    for native in (True, False):
        sh.set_state(psi0)
        ctx.reset_stats()
        for _ in range(3):
            sh.step(native=native)
        sh.step(backward=True, native=native)
        torch.cuda.synchronize()
        sh.check()
        results.append((sh.local_state(), sh.be.read(sh.X[0], N, sh.ncols_local), ctx.stats()["n_matvec"]))
    assert np.array_equal(results[0][0], results[1][0])          # state
    assert np.array_equal(results[0][1], results[1][1])          # ghost slots after the last exchange
    assert results[0][2] == results[1][2] == 4 * 31
    H = synth.to_scipy(rp, col, vals, N)
    wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
    ref = psi0.copy()
    for _ in range(3):
        qo.cheby(ref, H, 1.0, wrk)
    qo.cheby(ref, H, -1.0, wrk)
    assert np.linalg.norm(results[0][0] - ref) < 1e-10
    # the library's all-gather on its own
    a = L.State(ctx, data=np.arange(5, dtype=complex) + 1j)
    b = L.State(ctx, n=5)
    if sh.comm is not None:
        sh.comm.allgather(a, b, 5)
        ctx.sync()
        assert np.array_equal(b.numpy(), a.numpy())
    ctx.close()


def test_batch_split_gather_under_the_nccl_group(pg):
    """BatchSplitCheby.gather() under the PRODUCT process group (nccl = RCCL, no CPU backend: a host-tensor collective
    raises "No backend type associated with device type cpu" there): the panel is reassembled through device tensors.
    World 1 here (RCCL forms no multi-rank communicator on one device); the gloo branch is covered by
    tests/test_sharded_gloo.py::test_batch_split_gloo at world 2 and 3."""
    import torch
    from oracle import qp_oracle as qo
    import qprop_amd.lib as L
    import qprop_amd.sharded as sharded
    import qprop_amd.synth as synth
    assert str(pg.get_backend()).lower() == "nccl"
    N, batch = 2048, 8
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bs = sharded.BatchSplitCheby(ctx, rp, col, vals, N, batch, 20.0, -10.0, 1.0)
    states = np.stack([synth.random_state(N, seed=900 + s) for s in range(batch)], axis=1)
    bs.set_states(states)
    bs.step()
    got = bs.gather()                        # all_gather_into_tensor on cuda tensors
    assert got.shape == (N, batch) and np.array_equal(got, bs.local_states())
    H = synth.to_scipy(rp, col, vals, N)
    for s in range(batch):
        ref = qo.cheby(states[:, s].copy(), H, 1.0, qo.ChebyWrk(states[:, s], 20.0, -10.0, 1.0))
        assert np.linalg.norm(got[:, s] - ref) < 1e-10
    bs.close()
    ctx.close()
