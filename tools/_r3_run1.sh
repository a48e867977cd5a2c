#!/bin/bash
# round 3, run 1: where the Hermitian-packed fused term stands beyond the Infinity Cache + PMC diagnosis
mkdir -p gpurun_out/r3
for L in 21 22; do
  for wg in 4 8; do
    python - <<PY 2>&1 | grep -E "^hrb|^N="
import sys
sys.path.insert(0, '.')
import qprop_amd.lib as L
L.tuning_set("hrb_wg", $wg)
import runpy
sys.argv = ["kbench.py", "--log2n", "$L", "--formats", "hrb", "--variants", "7,15,31", "--rounds", "5", "--steps", "3"]
print("hrb_wg", $wg)
runpy.run_path("tools/kbench.py", run_name="__main__")
PY
  done
done > gpurun_out/r3/kbench_sizes.txt 2>&1
tools/pmc_diag.sh hrb22 tools/kbench.py --log2n 22 --formats hrb --variants 15 --rounds 2 --steps 2 > gpurun_out/r3/pmc_hrb22.log 2>&1
python tools/pmc_diag_summary.py hrb22 hrb_spmv_kernel > gpurun_out/r3/hrb_n22_pmc_raw.txt 2>&1
cat gpurun_out/r3/kbench_sizes.txt
cat gpurun_out/r3/hrb_n22_pmc_raw.txt
