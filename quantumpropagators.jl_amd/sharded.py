"""Row-partitioned Chebyshev ``prop_step!`` across the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Rank r owns
the contiguous CSR row block [r0, r1) and the matching slices of every vector.  After each
fused mat-vec term the freshly written slice of the term vector is exchanged so that the
next term can gather from it:

* ``allgather``  -- ``all_gather_into_tensor`` of the N/G-row slices, in place in the
  full-length buffer (what BASELINE.json's north_star names);
* ``halo``       -- only the column runs a rank actually reads are sent, as grouped
  point-to-point RCCL sends straight out of / into the full-length buffers (no packing).
  For a banded H this is 2 x 4096 rows per rank instead of the whole vector; for a
  scattered H it degenerates to the all-gather volume and ``auto`` picks ``allgather``.

Chebyshev needs no reductions, so there is no other collective on the path.  The buffer
rotation is the single-GPU one (engine.hip qp_cheby_step): two full-length buffers G0/G1
that alternate between "gathered v1" and "local v0 -> v2 in place", plus a local
accumulator.

The local compute goes through a *backend* object; the product backend is
:class:`HipBackend` (C ABI, fails loudly without a GPU).  CPU tests inject a NumPy backend
from ``tests/`` to exercise the partition / exchange logic under ``gloo``.
"""
from __future__ import annotations

import numpy as np

from . import lib as L


# ----------------------------------------------------------------------------------------
# index work: which remote columns does my row block read?  (bit-exact, host)
# ----------------------------------------------------------------------------------------

def needed_runs(col, r0, r1, bounds, gap_merge=2048):
    """Contiguous runs [(owner, lo, hi)] of global column indices outside [r0, r1) that
    the local rows reference; runs of the same owner closer than ``gap_merge`` are merged."""
    col = np.asarray(col)
    remote = np.unique(col[(col < r0) | (col >= r1)]).astype(np.int64)
    runs = []
    if len(remote) == 0:
        return runs
    owner = np.searchsorted(bounds, remote, side="right") - 1
    start = 0
    for i in range(1, len(remote) + 1):
        brk = (i == len(remote)) or (owner[i] != owner[start]) or (remote[i] - remote[i - 1] > gap_merge)
        if brk:
            runs.append((int(owner[start]), int(remote[start]), int(remote[i - 1]) + 1))
            start = i
    return runs


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

    def make_operator(self, rowptr, col, vals, nloc, N, fmt):
        return L.Operator(self.ctx, [L.Matrix(self.ctx, nloc, N, rowptr, col, vals)], 0, fmt)

    def view(self, t, lo, hi):
        return L.State(self.ctx, n=hi - lo, device_ptr=t.data_ptr() + 16 * lo, keepalive=t)

    def term(self, op, x, xoff, v0, vout, acc_in, acc_out, c, beta, a_prev, a, phase):
        L.cheby_term(op, x, xoff, v0, vout, acc_in, acc_out, c, beta, a_prev, a, phase)

    def write(self, t, lo, arr):
        arr = np.ascontiguousarray(arr, dtype=np.complex128)
        t[2 * lo: 2 * (lo + len(arr))].copy_(self.torch.from_numpy(arr.view(np.float64)))

    def read(self, t, lo, hi):
        return t[2 * lo: 2 * hi].cpu().numpy().view(np.complex128).copy()

    def copy(self, dst, dlo, src, slo, n):
        dst[2 * dlo: 2 * (dlo + n)].copy_(src[2 * slo: 2 * (slo + n)])


class ShardedCheby:
    """Chebyshev propagator for one row block of a row-partitioned H."""

    def __init__(self, ctx, rowptr, col, vals, N, r0, r1, Delta, E_min, dt, fmt=L.FMT_AUTO,
                 exchange="auto", group=None, backend=None, limit=1e-12, gap_merge=2048):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.N, self.r0, self.r1 = int(N), int(r0), int(r1)
        self.nloc = self.r1 - self.r0
        self.be = backend if backend is not None else HipBackend(ctx)
        self.Delta, self.E_min, self.dt, self.limit = float(Delta), float(E_min), float(dt), float(limit)
        self.coeffs = L.cheby_coeffs(Delta, dt, limit)
        if len(self.coeffs) < 2:
            raise L.QPAssertionError(4, "Need at least 2 Chebychev coefficients")

        # ---- partition bookkeeping (index work) ----
        mine = np.array([self.r0, self.r1], dtype=np.int64)
        allb = [None] * self.world
        dist.all_gather_object(allb, mine.tolist(), group=group)
        self.bounds = np.array([b[0] for b in allb] + [allb[-1][1]], dtype=np.int64)
        assert self.bounds[0] == 0 and self.bounds[-1] == self.N and np.all(np.diff(self.bounds) > 0), \
            "row blocks must tile [0, N) in rank order"
        self.equal_blocks = bool(np.all(np.diff(self.bounds) == self.nloc))
        self.recv_runs = needed_runs(col, self.r0, self.r1, self.bounds, gap_merge)
        all_runs = [None] * self.world
        dist.all_gather_object(all_runs, self.recv_runs, group=group)
        # what I must send: runs requested by rank d whose owner is me
        self.send_runs = [(d, lo, hi) for d in range(self.world) if d != self.rank
                          for (o, lo, hi) in all_runs[d] if o == self.rank]
        for (_, lo, hi) in self.send_runs:
            assert self.r0 <= lo < hi <= self.r1
        halo_rows = sum(hi - lo for r in all_runs for (_, lo, hi) in r)
        full_rows = self.world * (self.N - self.nloc) if self.world > 1 else 1
        self.halo_fraction = halo_rows / max(full_rows, 1)
        if exchange == "auto":
            exchange = "halo" if (self.halo_fraction < 0.5 or not self.equal_blocks) else "allgather"
        if exchange == "allgather" and not self.equal_blocks:
            raise ValueError("allgather exchange needs equal row blocks; use exchange='halo'")
        self.exchange = exchange

        # ---- device data ----
        self.op = self.be.make_operator(rowptr, col, vals, self.nloc, self.N, fmt)
        self.G = [self.be.zeros(self.N), self.be.zeros(self.N)]
        self.acc_t = self.be.zeros(self.nloc)
        self.Gfull = [self.be.view(g, 0, self.N) for g in self.G]
        self.Gloc = [self.be.view(g, self.r0, self.r1) for g in self.G]
        self.acc = self.be.view(self.acc_t, 0, self.nloc)
        self.n_exchanges = 0

    # the state lives in G[0][r0:r1]
    def set_state(self, psi_local):
        assert len(psi_local) == self.nloc
        self.be.write(self.G[0], self.r0, psi_local)

    def local_state(self):
        return self.be.read(self.G[0], self.r0, self.r1)

    def _exchange(self, k):
        """Make G[k] readable wherever the local rows gather from it."""
        if self.world == 1:
            return
        dist = self.dist
        g = self.G[k]
        if self.exchange == "allgather":
            dist.all_gather_into_tensor(g, g[2 * self.r0: 2 * self.r1], group=self.group)
        else:
            ops = []
            for (d, lo, hi) in self.send_runs:
                ops.append(dist.P2POp(dist.isend, g[2 * lo: 2 * hi], d, group=self.group))
            for (o, lo, hi) in self.recv_runs:
                ops.append(dist.P2POp(dist.irecv, g[2 * lo: 2 * hi], o, group=self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        self.n_exchanges += 1

    def step(self, backward=False):
        """One ``cheby!`` (src/cheby.jl:150-213) on the partitioned state."""
        a = self.coeffs
        Delta = self.Delta
        dt = -self.dt if backward else self.dt
        beta = Delta / 2 + self.E_min
        c = (-2j / Delta) if dt > 0 else (2j / Delta)
        phase = np.exp(-1j * beta * dt)
        nterms = len(a) - 1
        be, op, r0 = self.be, self.op, self.r0
        self._exchange(0)
        result_in_acc = True
        for m in range(1, nterms + 1):
            last = m == nterms
            xi, oi = (0, 1) if m % 2 == 1 else (1, 0)
            x, oloc = self.Gfull[xi], self.Gloc[oi]
            ph = phase if last else 1.0
            if m == 1:
                be.term(op, x, r0, None, None if last else oloc, None, self.acc, c, beta, a[0], a[1], ph)
            else:
                # the state buffer G[0] may be written only while it is not being gathered
                out = self.Gloc[0] if (last and xi == 1) else self.acc
                be.term(op, x, r0, oloc, None if last else oloc, self.acc, out, c, beta, 0.0, a[m], ph)
                result_in_acc = out is self.acc
            if not last:
                self._exchange(oi)
            if m == 1:
                c = 2 * c
        if result_in_acc:
            be.copy(self.G[0], self.r0, self.acc_t, 0, self.nloc)
