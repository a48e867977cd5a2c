#!/bin/bash
# rocprofv3 kernel stats of the Newton C3 bench, both orthogonalisation modes
set -u
TAG=${1:-r01}
MODES=${2:-"1 0"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_newton_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in $MODES; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mode$mode -- python3 $R/tools/bench_newton.py --format rbcsr --steps 10 --arnoldi-mode $mode > $OUT/mode$mode.log 2>&1
  tail -1 $OUT/mode$mode.log | cut -c100-400
  f=$(ls $OUT/mode$mode/*/*_kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cut -c1-60,150-400 "$f" | head -14; fi
done
