#!/usr/bin/env python3
"""Cost of the multi-GPU machinery measured on ONE GPU: the sharded driver at world = 1 with
a forced send set of 2 x 4096 rows (what a rank of the banded C2/C4 workload sends), with and
without the boundary/interior overlap, against the plain single-GPU step."""
import os, sys, time, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import qprop_amd.lib as L, qprop_amd.sharded as sharded, qprop_amd.synth as synth

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
N = 1 << int(os.environ.get("QP_LOG2N", "20"))
rp, col, vals = synth.hermitian_offsets_csr(N)
ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.tuning_set("hrb_walk", int(os.environ.get("QP_HRB_WALK", "1")))      # 0: the per-block kernels everywhere (A/B of the strip walk)
psi0 = synth.random_state(N)
send = np.concatenate([np.arange(0, 4096), np.arange(N - 4096, N)])
K = int(os.environ.get("QP_STEPS", "30"))
res = {}
for name, kw in (("plain (no exchange)", dict()), ("exchange, serial", dict(_debug_send_rows=send, overlap=False)),
                 ("exchange, overlap/events", dict(_debug_send_rows=send, overlap=True, mode=0)),
                 ("exchange, overlap/flag", dict(_debug_send_rows=send, overlap=True, mode=1)),
                 ("native, serial", dict(_debug_send_rows=send, overlap=False, native=True)),
                 ("native, overlap/flag", dict(_debug_send_rows=send, overlap=True, mode=1, native=True))):
    if os.environ.get("QP_ONLY") and os.environ["QP_ONLY"] != name:
        continue
    L.tuning_set("split_mode", kw.pop("mode", 1))
    sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", **kw)
    sh.set_state(psi0)
    for _ in range(3):
        sh.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        sh.step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    res[name] = (1e3 * t_host / K, 1e3 * t_all / K)
    print(f"{name:24s}: host enqueue {1e3*t_host/K:.3f} ms/step, wall {1e3*t_all/K:.3f} ms/step "
          f"({1e3*t_all/K/31*1e3:.1f} us/term), format={sh.op.format}, split={'yes' if sh.split else 'no'}", flush=True)
    del sh
dist.destroy_process_group()
