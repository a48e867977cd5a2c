#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python tools/_r3_walk_check.py > gpurun_out/r3/walk_check.txt 2>&1
tail -1 gpurun_out/r3/walk_check.txt | cut -c1-200
for L in 18 19 20; do
timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab walk_waves=512,768,1024,1280,1536,1792 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B" | cut -c1-120
timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab hrb_walk=0,1 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B" | cut -c1-120
done
for L in 21 22; do
timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab hrb_walk=0,1 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B" | cut -c1-120
done
timeout 900 python tools/kbench.py --log2n 20 --formats hrb --variants 15 --real --ab hrb_walk=0,1 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B" | cut -c1-120
