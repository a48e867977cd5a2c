"""End-to-end multi-rank test of the product HIP path on ONE GPU: 2 and 3 processes share GPU
0 (collective staged through the host with gloo -- see tests/multirank_gpu_worker.py).  This
file sorts first on purpose: the children must be started before this process initialises the
GPU (a GPU-initialised parent must not fork+exec on the test pool)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, timeout=240, **env_extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    # (a rank that dies takes the others' barrier down with it: show the rank that failed FIRST, i.e. not with
    # the "connection closed by peer" of a survivor)
    bad = [(p.returncode, out) for p, out in zip(procs, outs) if p.returncode != 0]
    bad.sort(key=lambda t: "Connection closed by peer" in t[1])
    assert not bad, "\n-----\n".join(f"rc {rc}: {out[-1500:]}" for rc, out in bad[:3])
    assert all("err=" in out for out in outs)
    return outs


@pytest.mark.parametrize("world,overlap,exchange,uneven", [
    (2, "1", "auto", "0"),
    (2, "0", "auto", "0"),
    (3, "1", "auto", "1"),
    (2, "1", "allgather", "0"),
])
def test_multirank_hip_path_one_gpu(world, overlap, exchange, uneven):
    _run(world, QP_OVERLAP=overlap, QP_EXCHANGE=exchange, QP_UNEVEN=uneven)


@pytest.mark.parametrize("world,overlap,exchange,uneven,p2p", [
    (2, "1", "auto", "0", "1"),        # neighbour lists (send/recv form of the exchange), overlap
    (3, "1", "auto", "1", "1"),        # three ranks, uneven blocks
    (3, "0", "auto", "1", "0"),        # all-gather form, serial schedule
    (2, "1", "allgather", "0", "0"),   # scattered H: the slice itself is the send buffer
])
def test_multirank_native_driver_one_gpu(world, overlap, exchange, uneven, p2p):
    """qp_sharded_cheby_step (the whole partitioned cheby! in one library call) with 2 and 3 ranks
    sharing the GPU: the exchange is handed back through qp_comm_create_callback and staged through
    the host, everything else -- term loop, two streams, fused pack, neighbour slots -- is the code
    that runs with RCCL on one GPU per rank."""
    outs = _run(world, QP_OVERLAP=overlap, QP_EXCHANGE=exchange, QP_UNEVEN=uneven, QP_NATIVE="1", QP_P2P=p2p)
    assert all("native=yes" in o for o in outs)


@pytest.mark.parametrize("world", [2, 3])
def test_multirank_newton_one_gpu(world):
    """Row-partitioned newton! (all-reduced Arnoldi inner products) with ranks sharing the GPU."""
    _run(world, QP_METHOD="newton")


def test_c4_full_size_eight_ranks_one_gpu():
    """BASELINE configs[3] at its size: N = 2^24 rows as 8 ranks x 2^21 rows sharing the one GPU, the library's
    one-call step with the exchange through the callback communicator (the same code that runs with RCCL on one
    GPU per rank): norms, forward / backward round trip, and the state after one step against the C oracle on
    2^18-row windows inside a rank, across a rank boundary and across the periodic wrap."""
    outs = _run(8, timeout=900, QP_METHOD="c4")
    assert all("c4 N=2^24" in o and "exchange=halo" in o and "M=8192" in o for o in outs)
    assert sum("err=0.000e+00" not in o for o in outs) >= 3      # the windows were compared on the ranks that own them


def test_c4_full_size_eight_ranks_allgather_form_one_gpu():
    """The same size in the ALL-GATHER form (the collective BASELINE's north_star names): a scattered H whose send list is every
    rank's whole 2^21-row slice, 32 MiB per rank and term through the exchange; the state after one step against the C
    oracle at full size (computed once, on rank 0), norm and forward / backward round trip.  (dt = 0.1, 11 coefficients:
    each term of this form moves 256 MB through the host-staged exchange.)"""
    outs = _run(8, timeout=1500, QP_METHOD="c4-allgather")
    assert all("c4-allgather N=2^24" in o and "exchange=allgather" in o and f"M={1 << 21}" in o for o in outs)


def test_bench_eight_ranks_full_c4_flow_one_gpu():
    """The COMPLETE `bench.py --gpus 8` flow at config C4's size (2^21 rows per rank, N = 2^24), as the driver launches it,
    with the 8 ranks sharing the test GPU: conservative pass, native set-up + self-check, both schedules' trial, the timed
    steps, the strong point -- so that the first run on a real 8-GPU node exercises nothing for the first time but RCCL
    itself.  Must finish well inside the driver's patience (asserted: < 10 min)."""
    import json
    import time
    world = 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    t0 = time.time()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", QP_BENCH_ONE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2",
                                       "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    took = time.time() - t0
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    assert took < 600, f"the full --gpus 8 flow took {took:.0f} s"
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    # the printed line is the compact one the driver parses: below 4 KB, no long strings, the gate keys
    assert len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["degraded"] is False and d["native_path"].startswith("ok") and d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["config"]["N_total"] == 1 << 24 and d["config"]["rows_per_gpu"] == 1 << 21
    par = d["config"]["parallelism"]
    # (the line reports the faster of the conservative and the native measurement; the native path's schedule and driver are named either way)
    assert "row-partitioned x8" in par and "exchange=halo" in par and "schedule=" in par and "driver=native" in par and len(par) <= 160
    assert d["roofline"]["frac"] > 0 and d["roofline"]["bound"] == "hbm" and d["value"] > 0
    assert d["strong_point"]["N_total"] == 1 << 20 and d["strong_point"]["prop_steps_per_s"] > 0
    # the collective north_star names has a measured point of its own next to the halo headline
    ag = d["allgather_form"]
    assert ag["us_per_term"] > 0 and ag["blocks_per_s"] > 0 and "error" not in ag
    assert d["prediction"]["fixed_n24_speedup_8gpu_halo"] > 6.0 > d["prediction"]["fixed_n24_speedup_8gpu_allgather"]
    # the complete record next to the script
    assert d["extras_file"] == "bench_extras_c4_gpus8.json"
    with open(os.path.join(ROOT, d["extras_file"])) as f:
        full = json.load(f)
    assert full["config"]["blocks_of_2^20_rows_per_step"] == 16.0 and "schedule=auto: overlap" in full["config"]["parallelism"] and "driver=native" in full["config"]["parallelism"]
    sp_ = full["strong_scaling_point"]
    assert sp_["N_total"] == 1 << 20 and sp_["rows_per_gpu"] == 1 << 17 and sp_["prop_steps_per_s"] > 0
    xm = full["exchange_model"]
    assert xm["rows_sent_per_rank_per_term"] == 8192 and xm["peers"] == 2
    pred = full["scaling_prediction"]         # the 1 / 2 / 4 / 8 table the first real run is read against
    assert [r["gpus"] for r in pred["fixed_problem_N_2^24"]] == [1, 2, 4, 8]
    assert pred["fixed_problem_N_2^24"][-1]["speedup_halo_overlap"] > 6.0 > pred["fixed_problem_N_2^24"][-1]["speedup_allgather"]
