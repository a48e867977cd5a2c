#!/usr/bin/env python3
"""Fuzz the row-partitioned Chebyshev step with 2 and 3 ranks sharing ONE GPU (collective staged
through the host; tests/multirank_gpu_worker.py with QP_FUZZ_SEED): random size, band structure,
uneven partition, exchange mode, overlapped / serial schedule, Python-driven or native (callback
communicator) driver, neighbour or all-gather form -- each against the NumPy oracle (1e-10).

    python tools/fuzz_sharded.py [n_cases] [first_seed]
Run it from a process that has not touched the GPU (children are separate processes)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(world, seed):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", QP_FUZZ_SEED=str(seed))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    ok = True
    lines = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            return False, ["TIMEOUT"]
        ok = ok and p.returncode == 0
        lines += [ln for ln in out.splitlines() if ln.startswith("rank ")] or [out[-800:]]
    return ok, lines


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for k in range(ncases):
        world = 2 + (k % 2)
        ok, lines = run(world, first + k)
        print(("ok  " if ok else "BAD ") + f"seed {first + k} world {world}: " + lines[0], flush=True)
        if not ok:
            bad += 1
            for ln in lines:
                print("    " + ln)
    print(f"{ncases} cases, {bad} bad")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
