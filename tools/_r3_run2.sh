#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python tools/_r3_walk_check.py > gpurun_out/r3/walk_check.txt 2>&1
tail -12 gpurun_out/r3/walk_check.txt
for L in 20 21 22; do
  timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab hrb_walk=0,1 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|norm"
done > gpurun_out/r3/kbench_walk.txt 2>&1
cat gpurun_out/r3/kbench_walk.txt
