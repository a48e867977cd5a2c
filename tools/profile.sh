#!/bin/bash
# rocprofv3 recipe for the headline bench (run on the GPU box through gpurun).
# usage: tools_profile.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-steps 0 --no-pmc --no-extras > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-pmc --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-pmc --no-extras > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-pmc --no-extras > $OUT/pmc_l2.log 2>&1
find $OUT -name "*.csv" | head -50
