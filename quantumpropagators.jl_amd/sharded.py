"""Row-partitioned Chebyshev ``prop_step!`` across the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Rank r owns
the contiguous CSR row block [r0, r1) and the matching slices of every vector.  The local
operator is stored in *local numbering*: columns [0, nloc) are the rank's own rows, columns
>= nloc are ghost slots, laid out as ``nloc + owner * M + position`` in one staging region
that a single RCCL all-gather fills after every fused mat-vec term:

    slab  = v_local[send_idx]                     (pack: the rows some other rank reads)
    all_gather_into_tensor(x[nloc:], slab)        (every rank's slab, M entries each)

``send_idx`` is computed from the column indices (bit-exact index work): for a banded H it
is 2 x 4096 rows per rank, so a term moves 128 KiB per rank instead of the whole vector;
when the rows read most of the vector (scattered H) it degenerates to the plain all-gather
of the N/G-row slices that BASELINE.json's north_star names (no pack, the slice itself is the
send buffer).  Chebyshev needs no reductions, so this is the only collective on the path.
In local numbering the square part of the block is still Hermitian, so the engine keeps it
Hermitian-packed (ghost columns are "upper" by construction).

The buffer rotation is the single-GPU one (engine_cheby.hip qp_cheby_step): two x buffers that
alternate between "gathered v1" and "local v0 -> v2 in place", plus a local accumulator.

The local compute goes through a *backend* object; the product backend is
:class:`HipBackend` (C ABI, fails loudly without a GPU).  CPU tests inject a NumPy backend
from ``tests/`` to exercise the partition / exchange logic under ``gloo``.
"""
from __future__ import annotations

import numpy as np

from . import lib as L


# ----------------------------------------------------------------------------------------
# index work (host, bit-exact)
# ----------------------------------------------------------------------------------------

def remote_columns(col, r0, r1):
    """Sorted unique global column indices outside [r0, r1) that the local rows read."""
    col = np.asarray(col)
    return np.unique(col[(col < r0) | (col >= r1)]).astype(np.int64)


def split_by_owner(remote, bounds):
    """{owner: sorted global columns} for the rank owning each remote column."""
    owner = np.searchsorted(bounds, remote, side="right") - 1
    return {int(o): remote[owner == o] for o in np.unique(owner)}


def remap_columns(col, r0, r1, bounds, send_lists, M):
    """Global -> local numbering: own columns c -> c - r0; a remote column c owned by o ->
    nloc + o*M + (position of c in o's send list)."""
    col = np.asarray(col, dtype=np.int64)
    nloc = r1 - r0
    out = np.empty(len(col), dtype=np.int64)
    local = (col >= r0) & (col < r1)
    out[local] = col[local] - r0
    rem = ~local
    if rem.any():
        c = col[rem]
        owner = np.searchsorted(bounds, c, side="right") - 1
        pos = np.empty(len(c), dtype=np.int64)
        for o in np.unique(owner):
            sel = owner == o
            sl = send_lists[int(o)]
            p = np.searchsorted(sl, c[sel])
            assert np.all(p < len(sl)) and np.array_equal(sl[p], c[sel]), "remote column missing from the owner's send list"
            pos[sel] = p
        out[rem] = nloc + owner * M + pos
    return out


class HipBackend:
    """Product backend: torch CUDA buffers (so torch.distributed can move them) viewed by
    libqprop_hip through raw device pointers; kernels run on torch's current stream."""

    def __init__(self, ctx):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device)

    def zeros(self, n):
        return self.torch.zeros(2 * n, dtype=self.torch.float64, device=self.device)

    def index(self, idx):
        return self.torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(self.device)

    def make_operator(self, rowptr, col, vals, nloc, ncols, fmt):
        return L.Operator(self.ctx, [L.Matrix(self.ctx, nloc, ncols, rowptr, col, vals)], 0, fmt)

    def view(self, t, lo, hi):
        return L.State(self.ctx, n=hi - lo, device_ptr=t.data_ptr() + 16 * lo, keepalive=t)

    def term(self, op, x, xoff, v0, vout, acc_in, acc_out, c, beta, a_prev, a, phase, defer=None):
        L.cheby_term(op, x, xoff, v0, vout, acc_in, acc_out, c, beta, a_prev, a, phase, defer)

    # ---- overlap of the exchange with compute (boundary rows on a side stream) ----
    def make_split(self, op, send_rows):
        if op.format not in (L.FMT_RBCSR, L.FMT_HRB):
            return None
        return L.Split(op, send_rows)

    def new_stream(self):
        # high priority: the few boundary workgroups (and the collective behind them) should
        # not queue behind the thousands of interior workgroups of the previous term
        return self.torch.cuda.Stream(device=self.device, priority=-1)

    def stream(self, s):
        return self.torch.cuda.stream(s)

    def join(self, s):
        cur = self.torch.cuda.current_stream(self.device)
        s.wait_stream(cur)
        cur.wait_stream(s)

    def current_stream(self):
        return self.torch.cuda.current_stream(self.device)

    def set_stream(self, s):
        self.torch.cuda.set_stream(s)

    def term_split(self, op, split, side, first, x, xoff, v0, vout, acc_in, acc_out, slab, c, beta, a_prev, a, phase,
                   defer=None):
        L.cheby_term_split(op, split, side.cuda_stream, first, x, xoff, v0, vout, acc_in, acc_out, slab, c, beta,
                           a_prev, a, phase, defer)

    def write(self, t, lo, arr):
        arr = np.ascontiguousarray(arr, dtype=np.complex128)
        t[2 * lo: 2 * (lo + len(arr))].copy_(self.torch.from_numpy(arr.view(np.float64)))

    def read(self, t, lo, hi):
        return t[2 * lo: 2 * hi].cpu().numpy().view(np.complex128).copy()

    # ---- Arnoldi / Newton building blocks ----
    def krylov(self, n, nvec):
        return L.Krylov(self.ctx, n, nvec)

    def mul(self, op, x, y):
        op.mul(x, y)


class ShardedCheby:
    """Chebyshev propagator for one row block of a row-partitioned H.

    ``rowptr, col, vals``: the local rows [r0, r1) in CSR with *global* column indices."""

    def __init__(self, ctx, rowptr, col, vals, N, r0, r1, Delta, E_min, dt, fmt=L.FMT_AUTO,
                 exchange="auto", group=None, backend=None, limit=1e-12, overlap=True,
                 host_staged=False, native=False, p2p="auto", _debug_send_rows=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        # host_staged: move the slabs through pinned host memory and a CPU collective (gloo).
        # Only for testing the full HIP path with several ranks sharing one GPU, where RCCL
        # refuses to form a communicator; never used by bench.py.
        self.host_staged = bool(host_staged)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.N, self.r0, self.r1 = int(N), int(r0), int(r1)
        self.nloc = nloc = self.r1 - self.r0
        self.be = backend if backend is not None else HipBackend(ctx)
        self.Delta, self.E_min, self.dt, self.limit = float(Delta), float(E_min), float(dt), float(limit)
        self.coeffs = L.cheby_coeffs(Delta, dt, limit)
        if len(self.coeffs) < 2:
            raise L.QPAssertionError(4, "Need at least 2 Chebychev coefficients")
        # which terms touch the Psi accumulator (every third one, folding the two before it)
        self.sched = L.acc_schedule(self.coeffs)
        if exchange not in ("auto", "halo", "allgather"):
            raise ValueError(f"unknown exchange mode {exchange!r}")

        # ---- partition bookkeeping ----
        allb = [None] * self.world
        dist.all_gather_object(allb, [self.r0, self.r1], group=group)
        self.bounds = bounds = np.array([b[0] for b in allb] + [allb[-1][1]], dtype=np.int64)
        assert bounds[0] == 0 and bounds[-1] == self.N and np.all(np.diff(bounds) > 0), \
            "row blocks must tile [0, N) in rank order"
        remote = remote_columns(col, self.r0, self.r1)
        frac = len(remote) / max(self.N - nloc, 1)
        fracs = [None] * self.world
        dist.all_gather_object(fracs, frac, group=group)
        self.halo_fraction = float(max(fracs))
        if exchange == "auto":
            exchange = "halo" if self.halo_fraction < 0.5 else "allgather"
        self.exchange = exchange

        # ---- send lists: which of my rows does any other rank read? ----
        self.send_to = self.recv_from = None       # neighbour lists (halo exchange only)
        if exchange == "halo":
            wanted = split_by_owner(remote, bounds)
            all_wanted = [None] * self.world
            dist.all_gather_object(all_wanted, wanted, group=group)
            send_lists = []
            for o in range(self.world):
                parts = [w[o] for w in all_wanted if o in w]
                send_lists.append(np.unique(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int64))
            self.recv_from = sorted(int(o) for o in wanted)
            self.send_to = sorted(r for r, w in enumerate(all_wanted) if self.rank in w)
        else:
            send_lists = [np.arange(bounds[o], bounds[o + 1], dtype=np.int64) for o in range(self.world)]
        if _debug_send_rows is not None:     # single-rank tests of the pack / overlap machinery
            send_lists = [np.asarray(_debug_send_rows, dtype=np.int64) + self.r0]
            self.send_to = self.recv_from = [0]
        self.send_lists = send_lists
        self.M = M = int(max(len(s) for s in send_lists)) if (self.world > 1 or _debug_send_rows is not None) else 0
        self.send_idx_host = send_lists[self.rank] - self.r0
        assert np.all((self.send_idx_host >= 0) & (self.send_idx_host < nloc))
        # the slice itself is the send buffer when every row is sent and blocks are equal
        self.direct_send = (exchange == "allgather" and len(self.send_idx_host) == nloc and M == nloc)
        self.ncols_local = nloc + self.world * M
        lcol = remap_columns(col, self.r0, self.r1, bounds, send_lists, M)

        # ---- device data ----
        be = self.be
        self.op = be.make_operator(rowptr, lcol, vals, nloc, self.ncols_local, fmt)
        self.X = [be.zeros(self.ncols_local), be.zeros(self.ncols_local)]
        self.acc_t = be.zeros(nloc)
        self.Xfull = [be.view(x, 0, self.ncols_local) for x in self.X]
        self.Xloc = [be.view(x, 0, nloc) for x in self.X]
        self.acc = be.view(self.acc_t, 0, nloc)
        self.exchanging = M > 0
        self.split = None
        if self.exchanging and not self.direct_send:
            self.slab = be.zeros(M)
            self.slab_state = be.view(self.slab, 0, M)
            idx = np.zeros(M, dtype=np.int64)            # padded with row 0 (never read by anyone)
            idx[: len(self.send_idx_host)] = self.send_idx_host
            self.send_idx = be.index(idx)
            # overlap: the few row blocks that feed or read the exchange run on a side stream,
            # so the all-gather of term m hides behind the interior rows of term m
            if overlap and hasattr(be, "make_split"):
                self.split = be.make_split(self.op, self.send_idx_host)
                if self.split is not None and self.split.n_interior == 0:
                    self.split = None
                if self.split is not None:
                    self.side = be.new_stream()
        self.n_exchanges = 0
        # native: the whole step is one library call, with the exchange on an RCCL communicator
        # the library owns (qp_sharded_cheby_step).  The torch.distributed group only carries
        # the communicator's id.  Needs the HIP backend and one GPU per rank.
        self.native = None
        self.p2p = False
        self.native_error = None
        if native and isinstance(be, HipBackend):
            # Every step below that can fail is followed by an agreement over the torch group, so
            # that either all ranks use the native driver or none does (a rank that raised while the
            # others entered a collective would leave them waiting).
            def agree(err):
                errs = [None] * self.world
                dist.all_gather_object(errs, None if err is None else str(err), group=group)
                bad = [e for e in errs if e is not None]
                return bad[0] if bad else None

            def exchange_id(uid):
                box = [uid]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                           group=group)
                if box[0] is None:
                    raise RuntimeError("rank 0 could not obtain an RCCL id")
                return box[0]
            comm, err = None, None
            if self.exchanging and self.host_staged:
                # testing with several ranks on one GPU: the library's step loop with the exchange
                # handed back to this process (host-staged through the torch group)
                comm = L.CallbackComm(ctx, self.rank, self.world, self._host_exchange)
            elif self.exchanging:
                try:
                    comm = L.Comm(ctx, self.rank, self.world, exchange_id, agree=agree)
                except Exception as exc:      # noqa: BLE001 -- reported, and the Python driver is used instead
                    err = exc
                err = agree(err)
            self.comm = comm if err is None else None
            # neighbour exchange when every rank talks to few others (banded H: 2), else all-gather
            npeers = [None] * self.world
            dist.all_gather_object(npeers, 0 if self.send_to is None else max(len(self.send_to), len(self.recv_from)),
                                   group=group)
            use_p2p = (self.exchanging and self.exchange == "halo" and not self.direct_send and
                       (p2p is True or (p2p == "auto" and max(npeers) <= max(2, self.world // 2))))
            stepper = None
            if err is None:
                try:
                    stepper = L.ShardedChebyStepper(
                        self.op, self.split, comm, self.Xfull[0], self.Xfull[1], self.acc,
                        None if (not self.exchanging or self.direct_send) else self.slab_state,
                        self.send_idx_host if (self.exchanging and not self.direct_send) else np.zeros(0, dtype=np.int64),
                        M, self.direct_send, self.send_to if use_p2p else None, self.recv_from if use_p2p else None)
                except Exception as exc:      # noqa: BLE001
                    err = exc
                err = agree(err)
            if err is None:
                self.native, self.p2p = stepper, bool(use_p2p)
            else:
                self.native_error = str(err)
                if stepper is not None:
                    stepper.close()
                if comm is not None:
                    comm.close()
                    self.comm = None
        # start the ranks aligned: a rank that is still building its operator must not keep
        # the others waiting inside their first collective
        dist.barrier(group=group)

    def _host_exchange(self, send_ptr, count, recv_ptr, send_to, recv_from, stream):
        """Exchange callback of the native driver in host-staged mode (qp_exchange_cb): every
        rank's slab through pinned host memory and a CPU all-gather; only the slabs this rank
        reads (``recv_from``; all for an all-gather) are written to its ghost slots."""
        torch = self.torch
        torch.cuda.synchronize()
        h_send = torch.from_numpy(L.State(self.be.ctx, n=count, device_ptr=send_ptr).numpy().view(np.float64))
        h_all = torch.empty(self.world * h_send.numel(), dtype=h_send.dtype)
        self.dist.all_gather_into_tensor(h_all, h_send, group=self.group)
        h_all = h_all.numpy().view(np.complex128).reshape(self.world, count)
        for o in (range(self.world) if recv_from is None else recv_from):
            L.State(self.be.ctx, n=count, device_ptr=recv_ptr + 16 * count * o).upload(h_all[o])
        torch.cuda.synchronize()

    def close(self):
        """Release the library handles now, in dependency order (stepper, communicator, split, operator),
        instead of whenever the garbage collector gets to them -- e.g. before the process group that the
        communicator was bootstrapped over is destroyed.  The object must not be used afterwards."""
        if getattr(self, "torch", None) is not None and isinstance(self.be, HipBackend):
            self.torch.cuda.synchronize()
        for name in ("native", "comm", "split", "op"):
            h = getattr(self, name, None)
            if h is not None and hasattr(h, "close"):
                h.close()
            setattr(self, name, None)

    def check(self):
        """Synchronise and verify that no in-launch wait of the overlapped schedule timed out."""
        if self.split is not None and hasattr(self.split, "check"):
            self.split.check()

    # the state lives in X[0][:nloc]
    def set_state(self, psi_local):
        assert len(psi_local) == self.nloc
        self.be.write(self.X[0], 0, psi_local)

    def local_state(self):
        return self.be.read(self.X[0], 0, self.nloc)

    def _exchange(self, k, packed=False):
        """Fill the ghost slots of X[k] with the other ranks' rows of the same vector.
        ``packed``: the slab already holds the send rows (fused pack of the term kernel)."""
        if not self.exchanging:
            return
        x = self.X[k]
        nloc = self.nloc
        if self.direct_send:
            send = x[: 2 * nloc]
        else:
            send = self.slab
            if not packed:
                self.torch.index_select(x[: 2 * nloc].view(-1, 2), 0, self.send_idx, out=send.view(-1, 2))
        if self.host_staged:
            h_send = send.cpu()                              # synchronises the current stream
            h_recv = self.torch.empty(self.world * h_send.numel(), dtype=h_send.dtype)
            self.dist.all_gather_into_tensor(h_recv, h_send, group=self.group)
            x[2 * nloc:].copy_(h_recv)
        else:
            self.dist.all_gather_into_tensor(x[2 * nloc:], send, group=self.group)
        self.n_exchanges += 1

    def step(self, backward=False, native=None):
        """One ``cheby!`` (src/cheby.jl:150-213) on the partitioned state.  ``native``:
        force (True) or forbid (False) the one-call library step; default: use it if built."""
        if native is None:
            native = self.native is not None
        if native:
            self.native.step(self.coeffs, self.Delta, self.E_min, -self.dt if backward else self.dt)
            self.n_exchanges += len(self.coeffs) - 1 if self.exchanging else 0
            return None
        if self.split is not None:
            return self._step_overlapped(backward)
        a = self.coeffs
        Delta = self.Delta
        dt = -self.dt if backward else self.dt
        beta = Delta / 2 + self.E_min
        c = (-2j / Delta) if dt > 0 else (2j / Delta)
        phase = np.exp(-1j * beta * dt)
        nterms = len(a) - 1
        be, op = self.be, self.op
        self._exchange(0)
        result_in_acc = True
        updated = False                 # has a term written the accumulator yet?
        for m in range(1, nterms + 1):
            last = m == nterms
            xi, oi = (0, 1) if m % 2 == 1 else (1, 0)
            x, oloc = self.Xfull[xi], self.Xloc[oi]
            ph = phase if last else 1.0
            d = self.sched[m - 1]
            acc_in = self.acc if (updated and not d.skip) else None
            a_prev = 0.0 if updated else a[0]
            if m == 1:
                be.term(op, x, 0, None, None if last else oloc, None, None if d.skip else self.acc, c, beta, a_prev,
                        a[1], ph, d)
            else:
                # the state buffer X[0] may be written only while it is not being gathered
                out = self.Xloc[0] if (last and xi == 1) else self.acc
                be.term(op, x, 0, oloc, None if last else oloc, acc_in, None if d.skip else out, c, beta, a_prev,
                        a[m], ph, d)
                result_in_acc = out is self.acc
            updated = updated or not d.skip
            if not last:
                self._exchange(oi)
            if m == 1:
                c = 2 * c
        if result_in_acc:
            self.X[0][: 2 * self.nloc].copy_(self.acc_t)

    def _step_overlapped(self, backward=False):
        """Same arithmetic as :meth:`step`; per term the boundary row blocks run on the side
        stream, immediately followed there by the all-gather of their packed output, while
        the interior row blocks (which read no ghost slot) run on the main stream."""
        a = self.coeffs
        Delta = self.Delta
        dt = -self.dt if backward else self.dt
        beta = Delta / 2 + self.E_min
        c = (-2j / Delta) if dt > 0 else (2j / Delta)
        phase = np.exp(-1j * beta * dt)
        nterms = len(a) - 1
        be, op, side = self.be, self.op, self.side
        be.join(side)
        # the side stream is the thread's current stream for the whole step, so that the
        # collectives order themselves against it without a per-term stream context switch;
        # the interior launches go to the context's (main) stream explicitly
        main = be.current_stream()
        be.set_stream(side)
        try:
            self._exchange(0)
            result_in_acc = True
            updated = False
            for m in range(1, nterms + 1):
                last = m == nterms
                xi, oi = (0, 1) if m % 2 == 1 else (1, 0)
                x, oloc = self.Xfull[xi], self.Xloc[oi]
                ph = phase if last else 1.0
                slab = None if last else self.slab_state
                d = self.sched[m - 1]
                acc_in = self.acc if (updated and not d.skip) else None
                a_prev = 0.0 if updated else a[0]
                if m == 1:
                    be.term_split(op, self.split, side, True, x, 0, None, None if last else oloc, None,
                                  None if d.skip else self.acc, slab, c, beta, a_prev, a[1], ph, d)
                else:
                    out = self.Xloc[0] if (last and xi == 1) else self.acc
                    be.term_split(op, self.split, side, False, x, 0, oloc, None if last else oloc, acc_in,
                                  None if d.skip else out, slab, c, beta, a_prev, a[m], ph, d)
                    result_in_acc = out is self.acc
                updated = updated or not d.skip
                if not last:
                    self._exchange(oi, packed=True)
                if m == 1:
                    c = 2 * c
        finally:
            be.set_stream(main)
        be.join(side)
        if result_in_acc:
            self.X[0][: 2 * self.nloc].copy_(self.acc_t)


class ShardedNewton:
    """Newton / restarted Arnoldi (``newton!``, src/newton.jl:246-385) for one row block of a
    row-partitioned generator.  Vectors are sharded like in :class:`ShardedCheby`; the Arnoldi
    inner products are partial sums over the local rows that are all-reduced (two small
    collectives per column: 2(j+1) inner products, then the 256 partial sums of the norm),
    the mat-vec input is completed by the same packed all-gather of the ghost slots, and the
    small dense algebra (Hessenberg eigenvalues, Leja points, Newton coefficients) is done
    redundantly on every rank by the library's host routines, so that all ranks take the same
    restart decisions.  Serial schedule (no overlap): at the sizes where a single GPU holds the
    problem this path is latency-bound; it exists for Hilbert spaces that need several GPUs."""

    def __init__(self, ctx, rowptr, col, vals, N, r0, r1, m_max=10, fmt=L.FMT_AUTO, exchange="auto", group=None,
                 backend=None, host_staged=False):
        # partition, send lists, local operator and ghost exchange are those of the Chebyshev driver
        self.base = ShardedCheby(ctx, rowptr, col, vals, N, r0, r1, 2.0, -1.0, 1.0, fmt=fmt, exchange=exchange,
                                 group=group, backend=backend, overlap=False, host_staged=host_staged)
        b = self.base
        self.torch, self.dist, self.group, self.be = b.torch, b.dist, group, b.be
        self.nloc, self.N = b.nloc, b.N
        if m_max <= 2:
            raise L.QPArgumentError(11, "Newton propagation requires m_max > 2")
        if m_max >= self.N:
            m_max = self.N - 1
            if m_max <= 2:
                raise L.QPArgumentError(11, "Newton propagation requires state dimension > 2")
        self.m_max = m_max
        be = self.be
        ld = m_max + 1
        self.q = be.krylov(self.nloc, ld)
        self.qv = [self.q.view(i) for i in range(ld)]
        self.v_t = be.zeros(self.nloc)
        self.v = be.view(self.v_t, 0, self.nloc)
        self.psi_t = be.zeros(self.nloc)
        self.psi = be.view(self.psi_t, 0, self.nloc)
        self.red_t = be.zeros(2 * ld)
        self.red = be.view(self.red_t, 0, 2 * ld)
        self.np_t = be.zeros(256)
        self.npart = be.view(self.np_t, 0, 256)
        self.hess_t = be.zeros(ld * ld)
        self.hcol = [be.view(self.hess_t, j * ld, (j + 1) * ld) for j in range(ld)]
        self.hn_t = be.zeros(2 * ld)
        self.hnorm = [be.view(self.hn_t, 2 * j, 2 * j + 2) for j in range(ld)]
        self.restarts = 0
        self.n_matvec = 0

    def set_state(self, psi_local):
        self.be.write(self.psi_t, 0, psi_local)

    def local_state(self):
        return self.be.read(self.psi_t, 0, self.nloc)

    def _allreduce(self, t):
        if self.base.world > 1:
            if self.base.host_staged:
                h = t.cpu()
                self.dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, group=self.group)

    def _global_norm(self, t):
        s = (t * t).sum().reshape(1)
        self._allreduce(s)
        return float(np.sqrt(float(s.cpu()[0])))

    def _arnoldi(self, m, dt, norm_min):
        """``arnoldi!(Hess, q, m, v, H, dt; extended=true)``  src/arnoldi.jl:60-100."""
        b, be, ld = self.base, self.be, self.m_max + 1
        self.hess_t.zero_()
        self.hn_t.zero_()
        self.qv[0].copy_from(self.v)
        for j in range(m):
            b.Xloc[0].copy_from(self.qv[j])       # q_j into the local part of the exchange buffer
            b._exchange(0)
            be.mul(b.op, b.Xfull[0], self.qv[j + 1])                 # :82
            self.q.multidot(j, self.red)
            self._allreduce(self.red_t[: 4 * (j + 1)])
            self.q.project(j, dt, self.red, self.hcol[j], self.npart)  # :84-87
            self._allreduce(self.np_t)
            self.q.normalize(j, dt, norm_min, self.npart, self.hnorm[j])   # :88-97
        self.n_matvec += m
        hh = be.read(self.hess_t, 0, ld * ld).reshape(ld, ld)       # hh[j] = column j
        hn = be.read(self.hn_t, 0, 2 * ld).reshape(ld, 2)
        Hess = np.zeros((ld, ld), dtype=np.complex128, order="F")
        m_eff = m
        for j in range(m):
            if hn[j, 1].real < norm_min:                              # dimensionality exhausted  :91-95
                m_eff = j + 1
                break
        for j in range(m_eff):
            Hess[: j + 1, j] = hh[j, : j + 1]
            Hess[j + 1, j] = hn[j, 0]
        return m_eff, Hess

    def step(self, dt, func=None, norm_min=1e-14, relerr=1e-12, max_restarts=50):
        be, q = self.be, self.q
        f = (lambda z: np.exp(-1j * z)) if func is None else func
        m = self.m_max
        a = np.zeros(10 * self.m_max + 1, dtype=np.complex128)
        leja = np.zeros(10 * self.m_max + 1, dtype=np.complex128)
        n_a = n_leja = 0
        s = 0
        self.v.copy_from(self.psi)                                       # :268
        beta = self._global_norm(self.v_t)                              # :271
        self.v.scal(1.0 / beta)
        while True:
            m, Hess = self._arnoldi(m, float(dt), norm_min)             # :277
            if m == 1 and s == 0:                                       # :289-295
                self.psi.scal(complex(f(beta * Hess[0, 0])))
                break
            ritz = L.hessenberg_eigvals(Hess, m, accumulate=True)       # :297
            if s == 0:
                radius = 1.2 * float(np.max(np.abs(ritz)))
            n_s = n_leja
            leja, n_leja = L.extend_leja(leja, n_leja, ritz.copy(), m)
            a, n_a = L.extend_newton_coeffs(a, n_a, leja, func, n_leja, radius)
            mp = m + 1
            Hm = Hess[:mp, :mp]
            R = np.zeros(mp, dtype=np.complex128)
            P = np.zeros(mp, dtype=np.complex128)
            R[0] = beta
            P[0] = a[n_s] * beta
            for k in range(1, m):                                       # :334-343
                R = (Hm @ R - leja[n_s + k - 1] * R) / radius
                P = P + a[n_s + k] * R
            q.combine(self.psi, s != 0, 1.0, 0, m, P[:m], self.npart)   # :346-352
            self._allreduce(self.np_t)
            norm_psi = float(np.sqrt(be.read(self.np_t, 0, 256).real.sum()))
            R = (Hm @ R - leja[n_s + m - 1] * R) / radius               # :356-362
            beta = float(np.linalg.norm(np.abs(R)))
            R = R * (1.0 / beta)
            q.combine(self.v, True, R[0], 1, m, R[1:mp])                # :363-367
            if beta * abs(a[n_a - 1]) / (1 + norm_psi) < relerr:        # :370
                break
            s += 1
            if s > max_restarts:
                raise L.QPAssertionError(6, f"newton!: s={s} exceeds max_restarts={max_restarts}")
        self.restarts = s
        return s


class BatchSplitCheby:
    """SURVEY 8e "Batched" / BASELINE configs[4]: a panel of ``batch`` states x N rows, the batch split over the ranks --
    rank r owns states [r b, (r + 1) b), b = batch / world, and its own copy of H.  A step is ``cheby!`` on every state:
    there is NO communication in it (no reductions in the Chebyshev recursion, src/cheby.jl:150-213, and no rows to
    exchange); ``gather()`` reassembles the panel for output.

    The product path is the HIP engine (``qp_cheby_step_batched`` on this rank's b-state panel); ``panel_backend`` lets the
    CPU (gloo) tests inject a NumPy stand-in with the same two methods (``make(...)``, ``step(dt)``/``read()``/``write()``)."""

    def __init__(self, ctx, rowptr, col, vals, N, batch, Delta, E_min, dt, fmt=L.FMT_AUTO, group=None, panel_backend=None,
                 rank=None, world=None):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.rank = rank if rank is not None else (self.dist.get_rank(group) if self.dist else 0)
        self.world = world if world is not None else (self.dist.get_world_size(group) if self.dist else 1)
        if batch % self.world:
            raise ValueError(f"{self.world} ranks do not divide the {batch}-state panel")
        self.N, self.batch, self.b = int(N), int(batch), int(batch) // self.world
        self.s0, self.s1 = self.rank * self.b, (self.rank + 1) * self.b
        self.dt = float(dt)
        if panel_backend is not None:
            self.impl = panel_backend.make(rowptr, col, vals, self.N, self.b, Delta, E_min, dt)
        else:
            self.impl = _HipPanel(ctx, rowptr, col, vals, self.N, self.b, Delta, E_min, dt, fmt)

    def set_states(self, states):
        """``states``: (N, batch) array of the WHOLE panel (every rank takes its columns) or (N, b) of this rank's share."""
        states = np.asarray(states, dtype=np.complex128)
        if states.shape == (self.N, self.batch):
            states = states[:, self.s0:self.s1]
        if states.shape != (self.N, self.b):
            raise ValueError(f"expected a ({self.N}, {self.batch}) or ({self.N}, {self.b}) panel, got {states.shape}")
        self.impl.write(np.ascontiguousarray(states))

    def step(self, backward=False):
        self.impl.step(-self.dt if backward else self.dt)

    def local_states(self):
        return self.impl.read()

    def gather(self):
        """The whole (N, batch) panel on every rank (output only, never part of a step).  The collective runs on tensors of the
        kind the group's backend moves: device tensors under nccl (= RCCL; the product group of bench.py has no CPU backend),
        host tensors under gloo."""
        loc = np.ascontiguousarray(self.local_states())
        if self.dist is None:
            return loc
        import torch
        if str(self.dist.get_backend(self.group)).lower() == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            send = torch.view_as_real(torch.from_numpy(loc)).to(dev)                    # (N, b, 2) fp64
            recv = torch.empty((self.world,) + tuple(send.shape), dtype=torch.float64, device=dev)
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
            parts = torch.view_as_complex(recv.cpu()).numpy()                           # (world, N, b)
            return np.concatenate(list(parts), axis=1)
        parts = [torch.empty((self.N, self.b), dtype=torch.complex128) for _ in range(self.world)]
        self.dist.all_gather(parts, torch.from_numpy(loc), group=self.group)
        return np.concatenate([p.numpy() for p in parts], axis=1)

    def close(self):
        if self.impl is not None:
            self.impl.close()
            self.impl = None


class _HipPanel:
    def __init__(self, ctx, rowptr, col, vals, N, b, Delta, E_min, dt, fmt):
        self.ctx, self.N, self.b = ctx, N, b
        self.M = L.Matrix(ctx, N, N, rowptr, col, vals)
        self.op = L.Operator(ctx, [self.M], 0, fmt)
        self.panel = L.State(ctx, n=N * b)
        self.wrk = L.ChebyWrk(ctx, N * b, Delta, E_min, dt)

    def write(self, states):
        self.panel.upload(states.reshape(-1))       # panel X[i * b + s]

    def read(self):
        return self.panel.numpy().reshape(self.N, self.b)

    def step(self, dt):
        L.cheby_batched(self.panel, self.op, dt, self.wrk, self.b)

    def close(self):
        for h in (self.panel, self.wrk, self.op, self.M):
            h.close()
