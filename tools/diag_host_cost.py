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
N = 1 << 16
rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1,2,3,4,16,32,48,64))
ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
send = np.concatenate([np.arange(0, 256), np.arange(N - 256, N)])
sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", _debug_send_rows=send)
sh.set_state(synth.random_state(N)); sh.step(); torch.cuda.synchronize()
K = 2000
x, oloc = sh.Xfull[0], sh.Xloc[1]
t0 = time.perf_counter()
for _ in range(K):
    sh.be.term_split(sh.op, sh.split, sh.side, False, x, 0, oloc, oloc, sh.acc, sh.acc, sh.slab_state, 1j, 0.0, 0.0, 0.5, 1.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"term_split host cost: {1e6*(t1-t0)/K:.1f} us")
t0 = time.perf_counter()
for _ in range(K):
    sh._exchange(1, packed=True)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"all_gather_into_tensor host cost: {1e6*(t1-t0)/K:.1f} us")
t0 = time.perf_counter()
for _ in range(K):
    L.cheby_term(sh.op, x, 0, oloc, oloc, sh.acc, sh.acc, 1j, 0.0, 0.0, 0.5, 1.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"cheby_term host cost: {1e6*(t1-t0)/K:.1f} us")
t0 = time.perf_counter()
for _ in range(K):
    torch.index_select(sh.X[0][: 2 * N].view(-1, 2), 0, sh.send_idx, out=sh.slab.view(-1, 2))
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"index_select host cost: {1e6*(t1-t0)/K:.1f} us")
dist.destroy_process_group()
