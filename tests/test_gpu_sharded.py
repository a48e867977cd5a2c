"""GPU test of the product multi-GPU driver (HipBackend + torch.distributed nccl=RCCL).
Only one GPU is available to the test box, so this runs world_size = 1: it covers the
torch-buffer <-> C-ABI plumbing, stream sharing and the local-numbering operator build;
the exchange logic itself is covered by tests/test_sharded_gloo.py (CPU, world 2 and 3)
and tests/test_gpu_parity.py::test_cheby_term_row_partition (two shards on one GPU)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_hip_backend_world1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sharded_gpu_worker.py")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "err=" in r.stdout
