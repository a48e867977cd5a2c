#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python tools/_r3_walk_check.py > gpurun_out/r3/walk_check.txt 2>&1
tail -3 gpurun_out/r3/walk_check.txt | cut -c1-200
for L in 18 19 20 21 22; do
  for kv in "hrb_walk=0,1" "walk_dbg=0,2"; do
    timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab $kv --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B"
  done
done > gpurun_out/r3/kbench_walk4.txt 2>&1
cut -c1-120 gpurun_out/r3/kbench_walk4.txt
timeout 900 python tools/kbench.py --log2n 20 --formats hrb --variants 15 --ab walk_waves=512,768,1024,1536 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B" | cut -c1-120
