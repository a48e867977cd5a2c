#!/bin/bash
# rocprofv3 recipe for any python command of this repo (run on the GPU box through gpurun):
# one kernel-trace + stats pass, then each PMC counter group in its own pass (the pool requires it).
# usage: tools/profile_cmd.sh <tag> <script.py> [args...]     -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_l2}
# then:  python tools/summarize_pmc.py <tag> profiles/rNN/<name> <kernel substring>
set -u
TAG=$1
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
SCRIPT=$R/$1
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $SCRIPT "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $SCRIPT "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $SCRIPT "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $SCRIPT "$@" > $OUT/pmc_l2.log 2>&1
tail -2 $OUT/trace.log | cut -c1-300
f=$(ls $OUT/trace/*/*_kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cut -c1-70,150-330 "$f" | head -8; fi
