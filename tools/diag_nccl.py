import faulthandler, os, sys, time
faulthandler.dump_traceback_later(45, exit=True)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
t=time.time(); print("init...", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
if mode == "eager":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("nccl")
print("init done", time.time()-t, flush=True)
out=[None]; dist.all_gather_object(out, [1,2]); print("all_gather_object ok", out, time.time()-t, flush=True)
x=torch.zeros(2*8, dtype=torch.float64, device="cuda"); y=torch.ones(8, dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(x[8:], y); torch.cuda.synchronize(); print("all_gather_into_tensor ok", x.tolist(), time.time()-t, flush=True)
dist.barrier(); print("barrier ok", time.time()-t, flush=True)
dist.destroy_process_group(); print("destroy ok", time.time()-t, flush=True)
