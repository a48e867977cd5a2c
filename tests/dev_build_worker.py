"""Worker of tests/test_00_multirank_gpu.py::test_split_wait_timeout_is_reported_not_hung: runs against the DEVELOPER flavour of
the library (QPROP_HIP_LIB = lib/libqprop_hip_dev.so, built by `make dev` with -DQP_DEVELOPER), the only one that accepts the
knob that makes the boundary launches of a split term skip their completion signal."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import qprop_amd.lib as L
    import qprop_amd.sharded as sharded
    import qprop_amd.synth as synth
    assert L.load().qp_developer_build() == 1, "this worker needs the developer flavour of the library"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    N = 8192
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=(1, 2, 3, 4, 16, 32, 48, 64))
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    send = np.concatenate([np.arange(0, 200), np.arange(N - 200, N)])
    try:
        ctx.tuning_set("split_mode", 1)
        ctx.tuning_set("split_spin_log2", 10)
        ctx.tuning_set("split_dbg", 1)
        sh = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", overlap=True,
                                  _debug_send_rows=send)
        assert sh.split is not None
        sh.set_state(synth.random_state(N))
        with pytest.raises(L.QPError) as ei:
            for _ in range(3):                 # the flag is raised inside the first step's launches and seen by a later call
                sh.step()
                torch.cuda.synchronize()
            sh.check()
        assert ei.value.status == L.QP_E_INTERNAL and "timed out" in str(ei.value)
        with pytest.raises(L.QPError):         # the split stays poisoned: its state is not valid
            sh.split.check()
        torch.cuda.synchronize()
    finally:
        ctx.tuning_set("split_dbg", 0)
        ctx.tuning_set("split_spin_log2", 28)
        ctx.tuning_set("split_mode", 2)
    # a fresh split on the same context works again
    sh2 = sharded.ShardedCheby(ctx, rp, col, vals, N, 0, N, 20.0, -10.0, 1.0, exchange="halo", overlap=True, _debug_send_rows=send)
    sh2.set_state(synth.random_state(N))
    sh2.step()
    sh2.check()
    assert abs(np.linalg.norm(sh2.local_state()) - 1.0) < 1e-11
    ctx.close()
    dist.destroy_process_group()
    print("dev-build time-out test ok")


if __name__ == "__main__":
    main()
