#!/bin/bash
# every fuzzer and the overlap stress against the library as built (gpurun -- 'bash tools/run_fuzzers.sh'); outputs under gpurun_out/fz/
mkdir -p gpurun_out/fz
timeout 900 python tools/fuzz_sharded.py 40 500 > gpurun_out/fz/fuzz_sharded.txt 2>&1; echo "fuzz_sharded rc $?"; tail -1 gpurun_out/fz/fuzz_sharded.txt
timeout 1200 python tools/fuzz_parity.py 1500 57 > gpurun_out/fz/fuzz_parity.txt 2>&1; echo "fuzz_parity rc $?"; tail -1 gpurun_out/fz/fuzz_parity.txt
timeout 900 python tools/fuzz_newton_large.py 200 55 > gpurun_out/fz/fuzz_newton_large.txt 2>&1; echo "fuzz_newton_large rc $?"; tail -1 gpurun_out/fz/fuzz_newton_large.txt
timeout 900 python tools/stress_overlap.py > gpurun_out/fz/stress_overlap.txt 2>&1; echo "stress_overlap rc $?"
timeout 1500 python tools/fuzz_walk.py 500 511 > gpurun_out/fz/fuzz_walk.txt 2>&1; tail -1 gpurun_out/fz/fuzz_walk.txt
timeout 1500 python tools/fuzz_walk.py 500 531 > gpurun_out/fz/fuzz_walk_long_pair.txt 2>&1; tail -1 gpurun_out/fz/fuzz_walk_long_pair.txt
timeout 900 python tools/fuzz_liouville.py > gpurun_out/fz/fuzz_liouville.txt 2>&1; tail -1 gpurun_out/fz/fuzz_liouville.txt
timeout 900 python tools/fuzz_dense.py 400 53 > gpurun_out/fz/fuzz_dense.txt 2>&1; tail -1 gpurun_out/fz/fuzz_dense.txt
timeout 900 python tools/fuzz_colblock.py 300 55 > gpurun_out/fz/fuzz_colblock.txt 2>&1; tail -1 gpurun_out/fz/fuzz_colblock.txt
timeout 1500 python tools/fuzz_walk.py 400 5123 4 > gpurun_out/fz/fuzz_walk_diagonals_and_long_pairs.txt 2>&1; tail -1 gpurun_out/fz/fuzz_walk_diagonals_and_long_pairs.txt
timeout 900 python tools/fuzz_pauli.py 300 71 > gpurun_out/fz/fuzz_pauli.txt 2>&1; tail -1 gpurun_out/fz/fuzz_pauli.txt
timeout 1200 python tools/fuzz_spmm_tiles.py 300 91 > gpurun_out/fz/fuzz_spmm_tiles.txt 2>&1; tail -1 gpurun_out/fz/fuzz_spmm_tiles.txt
