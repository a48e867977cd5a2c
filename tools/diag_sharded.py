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
N = 4096
rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
psi0 = synth.random_state(N)
op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
wrk = L.ChebyWrk(ctx, N, 20.0, -10.0, 1.0)
ref = L.State(ctx, data=psi0)
L.cheby(ref, op, 1.0, wrk)
r1 = ref.numpy()
for ex in ("auto", "allgather"):
    sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange=ex)
    print(ex, "fmt", sh.op.format, "M", sh.M, "ncols_local", sh.ncols_local, "exchanging", sh.exchanging, "direct", sh.direct_send, "split", sh.split)
    sh.set_state(psi0)
    sh.step()
    torch.cuda.synchronize()
    out = sh.local_state()
    print("  err vs single-GPU step:", np.linalg.norm(out - r1), "norm", np.linalg.norm(out))
    r_, c_, v_ = sh.op.get_csr()
    print("  op roundtrip cols equal:", np.array_equal(c_, col), "vals:", np.array_equal(v_, vals))
dist.destroy_process_group()
