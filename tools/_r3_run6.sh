#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python tools/_r3_walk_check.py > gpurun_out/r3/walk_check.txt 2>&1
tail -1 gpurun_out/r3/walk_check.txt | cut -c1-200
for L in 20 22; do
  for dbg in 0 1; do
  python - <<PY 2>&1 | grep -E "^hrb|^N=|A/B|dbg" | cut -c1-120
import sys
sys.path.insert(0, '.')
import qprop_amd.lib as L
L.tuning_set("walk_dbg", $dbg)
import runpy
print("walk_dbg", $dbg)
sys.argv = ["kbench.py", "--log2n", "$L", "--formats", "hrb", "--variants", "15", "--ab", "walk_edge_steps=1,2,3,4,5", "--rounds", "5", "--steps", "3"]
runpy.run_path("tools/kbench.py", run_name="__main__")
PY
  done
done > gpurun_out/r3/kbench_walk5.txt 2>&1
cat gpurun_out/r3/kbench_walk5.txt
for L in 18 19 20; do
timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab walk_waves=-1,512,1024,2048 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B" | cut -c1-120
timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab hrb_walk=0,1 --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B" | cut -c1-120
done
