#!/usr/bin/env python3
"""Race hunt for the overlapped multi-GPU schedule on one GPU (world = 1, forced send set):
many steps of every schedule -- Python-driven (events / in-launch counter) and the native
one-call driver (all-gather, neighbour send/recv, serial) -- must stay bit-identical to the
serial Python-driven schedule, for several sizes (different kernel durations -> different
interleavings of the two streams)."""
import os, sys, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import qprop_amd.lib as L, qprop_amd.sharded as sharded, qprop_amd.synth as synth

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
bad = 0
for log2n, steps in ((14, 300), (16, 300), (18, 200), (20, 100)):
    N = 1 << log2n
    offs = synth.BANDED_OFFSETS if log2n >= 16 else (1, 2, 3, 4, 16, 32, 48, 64)
    w = max(offs)
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offs)
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    send = np.concatenate([np.arange(0, w), np.arange(N - w, N)])
    psi0 = synth.random_state(N)
    outs = {}
    for name, kw, mode in (("serial", dict(overlap=False), 1), ("flag", dict(overlap=True), 1), ("events", dict(overlap=True), 0),
                           ("native", dict(overlap=True, native=True, p2p=False), 1),
                           ("native-p2p", dict(overlap=True, native=True, p2p=True), 1),
                           ("native-serial", dict(overlap=False, native=True), 1)):
        L.tuning_set("split_mode", mode)
        sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", _debug_send_rows=send, **kw)
        sh.set_state(psi0)
        for k in range(steps):
            sh.step(backward=(k % 7 == 3))
        torch.cuda.synchronize()
        sh.check()
        outs[name] = (sh.local_state(), sh.be.read(sh.X[0], N, N + len(send)))
        del sh
    for name in ("flag", "events", "native", "native-p2p", "native-serial"):
        same = np.array_equal(outs[name][0], outs["serial"][0]) and np.array_equal(outs[name][1], outs["serial"][1])
        print(f"N=2^{log2n} steps={steps} {name:13s} == serial: {same}  norm={np.linalg.norm(outs[name][0]):.15f}", flush=True)
        bad += 0 if same else 1
    L.tuning_set("split_mode", 2)
    ctx.close()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
