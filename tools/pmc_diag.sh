#!/bin/bash
# PMC diagnosis of one kernel: a kernel-trace + stats pass, then one `rocprofv3 --pmc` pass per counter PAIR
# (the pool requires PMC passes without the tracing domains; a pass with too many counters of one block
# aborts), every pass under `timeout`, the program directly after `--`.
# usage: tools/pmc_diag.sh <tag> <script.py> [args...]      -> gpurun_out/pmc_<tag>/<pair>/
# then:  python tools/pmc_diag_summary.py <tag> <kernel substring>  > profiles/rNN/<name>.txt
set -u
TAG=$1
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
SCRIPT=$R/$1
shift
cd /tmp && export TMPDIR=/tmp
PAIRS=${QP_PMC_PAIRS:-"GRBM_GUI_ACTIVE,SQ_WAVES SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY,SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM,SQ_INSTS_VMEM_WR TCP_PENDING_STALL_CYCLES_sum,TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum,TCP_TCC_WRITE_REQ_sum TCC_HIT_sum,TCC_MISS_sum TCC_EA0_RDREQ_sum,TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum,TCC_TAG_STALL_sum TA_TA_BUSY_sum,TA_ADDR_STALLED_BY_TC_CYCLES_sum FETCH_SIZE WRITE_SIZE"}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $SCRIPT "$@" > $OUT/trace.log 2>&1
for p in $PAIRS; do
  timeout 600 rocprofv3 --pmc ${p//,/ } --kernel-trace --output-format csv -d $OUT/$p -- python3 $SCRIPT "$@" > $OUT/$p.log 2>&1
  echo "pass $p rc $?"
done
# keep what travels back small: the counter CSVs and the stats only
find $OUT -name "*kernel_trace.csv" -size +2M -delete
