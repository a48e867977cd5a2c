#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python tools/_r3_walk_check.py > gpurun_out/r3/walk_check.txt 2>&1
tail -8 gpurun_out/r3/walk_check.txt
for L in 19 20 21 22; do
  for kv in "walk_waves=-1,1024,2048" "walk_nt=0,1,3,7,4"; do
    timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab $kv --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B"
  done
done > gpurun_out/r3/kbench_walk3.txt 2>&1
cat gpurun_out/r3/kbench_walk3.txt
