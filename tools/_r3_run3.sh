#!/bin/bash
mkdir -p gpurun_out/r3
for L in 20 22; do
  for kv in "walk_waves=1024,2048,4096" "walk_dbg=0,1,2"; do
    timeout 900 python tools/kbench.py --log2n $L --formats hrb --variants 15 --ab $kv --rounds 5 --steps 3 2>&1 | grep -E "^hrb|^N=|A/B"
  done
done > gpurun_out/r3/kbench_walk2.txt 2>&1
cat gpurun_out/r3/kbench_walk2.txt
QP_PMC_PAIRS="SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY,SQ_INSTS_VMEM_RD TCP_PENDING_STALL_CYCLES_sum,TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum,TCP_TCC_WRITE_REQ_sum TCC_HIT_sum,TCC_MISS_sum TCC_EA0_RDREQ_sum,TCC_EA0_RDREQ_LEVEL_sum TA_TA_BUSY_sum,TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE,SQ_WAVES FETCH_SIZE WRITE_SIZE" tools/pmc_diag.sh walk22 tools/kbench.py --log2n 22 --formats hrb --variants 15 --rounds 2 --steps 2 > gpurun_out/r3/pmc_walk22.log 2>&1
python tools/pmc_diag_summary.py walk22 hrb_walk_kernel > gpurun_out/r3/walk_n22_pmc_raw.txt 2>&1
cat gpurun_out/r3/walk_n22_pmc_raw.txt
