"""Multi-process CPU test (gloo, world_size 2 and 3) of the row-partitioned Chebyshev
driver: partition bookkeeping, halo-run computation, both exchange modes, forward and
backward steps, uneven row blocks.  The local compute is a NumPy backend (the GPU kernels
are covered by tests/test_gpu_parity.py::test_cheby_term_row_partition); the result must
equal the oracle's unsharded cheby! to rounding."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, exchange, uneven, offsets, N, q, overlap=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from np_backend import NumpyBackend
        from oracle import qp_oracle as qo
        import qprop_amd.sharded as sharded
        import qprop_amd.synth as synth
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets)
        bounds = qo.partition_rows(rp, world)
        if uneven:
            bounds = bounds.copy()
            bounds[1:-1] += 37
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        lrp = rp[r0:r1 + 1] - rp[r0]
        lcol, lvals = col[rp[r0]:rp[r1]], vals[rp[r0]:rp[r1]]
        sh = sharded.ShardedCheby(None, lrp, lcol, lvals, N, r0, r1, 20.0, -10.0, 1.0, exchange=exchange,
                                  backend=NumpyBackend(), overlap=overlap)
        # neighbour lists of the halo exchange (what the native driver's ncclSend / ncclRecv
        # pairs are built from): o sends to r exactly when r receives from o, and a rank
        # receives from every owner of a ghost column it reads
        if sh.exchange == "halo":
            lists = [None] * world
            dist.all_gather_object(lists, (sh.send_to, sh.recv_from))
            for r_, (st, rf) in enumerate(lists):
                assert all(r_ in lists[o][0] for o in rf) and all(r_ in lists[o][1] for o in st)
            ghosts = lcol[(lcol < r0) | (lcol >= r1)]
            owners = set((np.searchsorted(bounds, ghosts, side="right") - 1).tolist())
            assert owners == set(sh.recv_from)
        else:
            assert sh.send_to is None and sh.recv_from is None
        psi0 = synth.random_state(N)
        sh.set_state(psi0[r0:r1])
        sh.step()
        sh.step()
        sh.step(backward=True)
        out = sh.local_state()
        # oracle, unsharded
        H = synth.to_scipy(rp, col, vals, N)
        wrk = qo.ChebyWrk(psi0, 20.0, -10.0, 1.0)
        ref = psi0.copy()
        qo.cheby(ref, H, 1.0, wrk)
        qo.cheby(ref, H, 1.0, wrk)
        qo.cheby(ref, H, -1.0, wrk)
        err = float(np.linalg.norm(out - ref[r0:r1]))
        q.put((rank, err, sh.exchange, sh.halo_fraction, sh.M, len(sh.send_idx_host), sh.n_exchanges,
               sh.split is not None))
        sh.close()                      # explicit release before the process group goes (idempotent)
        sh.close()
        assert sh.op is None and sh.split is None and sh.native is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange,uneven,offsets,overlap", [
    (2, "allgather", False, (1, 2, 3, 4, 16, 32, 48, 64), True),
    (2, "halo", False, (1, 2, 3, 4, 16, 32, 48, 64), True),
    (2, "halo", False, (1, 2, 3, 4, 16, 32, 48, 64), False),
    (2, "auto", True, (1, 2, 3, 4, 16, 32, 48, 64), True),
    (3, "auto", False, (1, 2, 3, 4, 16, 32, 48, 64), True),
    (2, "auto", False, (5, 77, 211, 333, 401, 467, 489, 499), True),      # scattered -> allgather
])
def test_sharded_cheby_matches_oracle(world, exchange, uneven, offsets, overlap):
    N = 1536
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, exchange, uneven, offsets, N, q, overlap))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    coeffs_terms = 31
    for rank, err, used, frac, M, nsend, nex, split in res:
        assert split == (overlap and used == "halo")        # overlapped path exercised where it applies
        assert err < 1e-12, (rank, err)
        assert nex == 3 * coeffs_terms       # one exchange per mat-vec: psi + 30 term vectors per step
        if exchange != "auto":
            assert used == exchange
    if offsets[-1] == 64 and exchange == "auto":
        # banded: every rank sends only its 2 x 64 boundary rows
        assert all(r[2] == "halo" and r[3] < 0.5 and r[5] <= 2 * 64 for r in res)
    if offsets[-1] == 499:
        assert all(r[2] == "allgather" for r in res)


def test_index_work():
    """Local numbering: own columns first, ghost slot = nloc + owner*M + position."""
    import qprop_amd.sharded as sharded
    bounds = np.array([0, 100, 200, 300])
    col = np.array([95, 99, 100, 150, 199, 200, 201, 204, 290, 299, 3])
    remote = sharded.remote_columns(col, 100, 200)
    assert list(remote) == [3, 95, 99, 200, 201, 204, 290, 299]
    by = sharded.split_by_owner(remote, bounds)
    assert sorted(by) == [0, 2] and list(by[0]) == [3, 95, 99] and list(by[2]) == [200, 201, 204, 290, 299]
    send_lists = [np.array([3, 50, 95, 99]), np.array([100, 199]), np.array([200, 201, 204, 290, 299])]
    M = 5
    lc = sharded.remap_columns(col, 100, 200, bounds, send_lists, M)
    assert list(lc) == [100 + 2, 100 + 3, 0, 50, 99, 100 + 10 + 0, 100 + 10 + 1, 100 + 10 + 2, 100 + 10 + 3,
                        100 + 10 + 4, 100 + 0]
    with pytest.raises(AssertionError):
        sharded.remap_columns(np.array([7]), 100, 200, bounds, send_lists, M)


def _newton_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from np_backend import NumpyBackend
        from oracle import qp_oracle as qo
        import qprop_amd.sharded as sharded
        import qprop_amd.synth as synth
        Lm = synth.liouvillian_tridiag(24)                       # N = 576, non-Hermitian
        N = Lm.shape[0]
        bounds = qo.partition_rows(Lm.indptr.astype(np.int64), world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        lrp = Lm.indptr[r0:r1 + 1].astype(np.int64) - Lm.indptr[r0]
        lcol, lvals = Lm.indices[Lm.indptr[r0]:Lm.indptr[r1]], Lm.data[Lm.indptr[r0]:Lm.indptr[r1]]
        sn = sharded.ShardedNewton(None, lrp, lcol, lvals, N, r0, r1, m_max=12, backend=NumpyBackend())
        rho0 = synth.random_state(N)
        sn.set_state(rho0[r0:r1])
        restarts = [sn.step(0.4), sn.step(0.4), sn.step(-0.4)]
        out = sn.local_state()
        ref = rho0.copy()
        owrk = qo.NewtonWrk(ref, m_max=12)
        oracle_restarts = []
        for dt in (0.4, 0.4, -0.4):
            qo.newton(ref, Lm, dt, owrk)
            oracle_restarts.append(owrk.restarts)
        q.put((rank, float(np.linalg.norm(out - ref[r0:r1])), restarts, oracle_restarts, sn.base.exchange))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_newton_matches_oracle(world):
    """Row-partitioned newton!: all-reduced Arnoldi inner products, replicated small dense algebra."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_newton_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    for rank, err, restarts, oracle_restarts, exch in res:
        assert err < 1e-10, (rank, err)
        assert all(abs(a - b) <= 1 for a, b in zip(restarts, oracle_restarts))
    assert len({tuple(r[2]) for r in res}) == 1          # every rank took the same restart decisions


def _batch_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from np_backend import NumpyPanelBackend
        import qprop_amd.sharded as sharded
        import qprop_amd.synth as synth
        N, batch = 512, 12
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 64, 128))
        states = np.stack([synth.random_state(N, seed=500 + s) for s in range(batch)], axis=1)
        bs = sharded.BatchSplitCheby(None, rp, col, vals, N, batch, 20.0, -10.0, 0.5, panel_backend=NumpyPanelBackend())
        assert (bs.s0, bs.s1) == (rank * batch // world, (rank + 1) * batch // world)
        bs.set_states(states)
        bs.step()
        bs.step()
        bs.step(backward=True)
        full = bs.gather()
        q.put((rank, full, bs.b))
        bs.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_batch_split_gloo(world):
    """SURVEY 8e "Batched" (BASELINE configs[4]): the panel's states split over the ranks, H replicated, no communication in
    a step; the gathered panel is the same on every rank and equals the oracle's cheby! of every state."""
    from oracle import qp_oracle as qo
    import qprop_amd.synth as synth
    port = _free_port()
    ctx_ = mp.get_context("spawn")
    q = ctx_.Queue()
    procs = [ctx_.Process(target=_batch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, batch = 512, 12
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 64, 128))
    H = synth.to_scipy(rp, col, vals, N)
    ref = np.stack([synth.random_state(N, seed=500 + s) for s in range(batch)], axis=1)
    for s_ in range(batch):
        psi = ref[:, s_].copy()
        w = qo.ChebyWrk(psi, 20.0, -10.0, 0.5)
        qo.cheby(psi, H, 0.5, w)
        qo.cheby(psi, H, 0.5, w)
        qo.cheby(psi, H, -0.5, w)
        ref[:, s_] = psi
    for rank, full, b in res:
        assert b == batch // world and full.shape == (N, batch)
        assert np.max(np.linalg.norm(full - ref, axis=0)) < 1e-12


def test_batch_split_rejects_uneven_split():
    import qprop_amd.sharded as sharded
    from np_backend import NumpyPanelBackend
    import qprop_amd.synth as synth
    rp, col, vals = synth.hermitian_offsets_csr(256, offsets=(1, 2))
    with pytest.raises(ValueError):
        sharded.BatchSplitCheby(None, rp, col, vals, 256, 10, 20.0, -10.0, 0.5, panel_backend=NumpyPanelBackend(), rank=0, world=4)


class _FakeCommLib:
    """Stand-in for the qp_comm_* entry points (the real ones need a GPU and librccl): records the calls."""

    def __init__(self, log):
        self.log = log

    def qp_comm_prepare(self, ctx, path, rank, world, out):
        self.log.append("prepare")
        out._obj.value = 0x1000          # a non-NULL handle
        return 0

    def qp_comm_unique_id(self, path, buf):
        self.log.append("unique_id")
        buf.raw = bytes(range(128))
        return 0

    def qp_comm_connect(self, h, uid):
        self.log.append("connect")
        return 0

    def qp_comm_destroy(self, h):
        self.log.append("destroy")
        return 0

    def qp_last_error(self):
        return b""


class _FakeCtx:
    def __init__(self, log):
        self.lib = _FakeCommLib(log)
        self._h = None

    def _adopt(self, child):
        pass


def _comm_worker(rank, world, port, q, die_rank):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))
    import qprop_amd.lib as L
    log = []

    def exchange_id(uid):
        if rank == die_rank:
            os._exit(17)        # this rank dies between qp_comm_prepare and qp_comm_connect
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def agree(err):
        errs = [None] * world
        dist.all_gather_object(errs, err)
        bad = [e for e in errs if e is not None]
        return bad[0] if bad else None

    try:
        L.Comm(_FakeCtx(log), rank, world, exchange_id, lib_path="librccl-not-loaded-here.so", agree=agree)
        q.put((rank, "connected", log))
    except BaseException as e:      # noqa: BLE001
        q.put((rank, "error:" + type(e).__name__, log))
    q.close()
    q.join_thread()                 # flush the result before leaving without interpreter shutdown:
    os._exit(0)                     # (the process group of a world that lost a rank cannot be torn down collectively)


@pytest.mark.parametrize("die_rank", [None, 1, 0])
def test_comm_setup_survives_a_rank_that_dies_between_prepare_and_connect(die_rank):
    """Two-phase set-up of the library's RCCL communicator (qprop_amd.lib.Comm): local preparation, then the ranks
    exchange rank 0's id and AGREE, and only then enter the collective connect.  If a rank dies in between, the others'
    exchange / agreement over the torch group fails (bounded by the group's time-out) and they raise with the handle
    released -- none of them enters qp_comm_connect, where it would block for good.  (CPU: the entry points are a fake
    that records calls; the protocol under test is the host side.)"""
    world = 3
    port = _free_port()
    ctx_ = mp.get_context("spawn")
    q = ctx_.Queue()
    procs = [ctx_.Process(target=_comm_worker, args=(r, world, port, q, die_rank)) for r in range(world)]
    for p in procs:
        p.start()
    alive = world - (0 if die_rank is None else 1)
    res = {}
    for _ in range(alive):
        rank, what, log = q.get(timeout=120)
        res[rank] = (what, log)
    for p in procs:
        p.join(timeout=60)
    if die_rank is None:
        assert all(what == "connected" and "connect" in log for what, log in res.values()), res
        return
    assert procs[die_rank].exitcode == 17 and die_rank not in res
    for rank, (what, log) in res.items():
        assert what.startswith("error:"), (rank, what)
        assert "connect" not in log and log[0] == "prepare" and log[-1] == "destroy", (rank, log)
