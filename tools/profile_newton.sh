#!/bin/bash
# rocprofv3 kernel stats of the Newton C3 bench, both orthogonalisation modes
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_newton_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mode$mode -- python3 $R/tools/bench_newton.py --format rbcsr --steps 10 --arnoldi-mode $mode > $OUT/mode$mode.log 2>&1
  tail -1 $OUT/mode$mode.log | cut -c100-400
  cat $OUT/mode$mode/*/*_kernel_stats.csv | cut -c1-60,150-400 | head -14
done
