#!/bin/bash
# usage (GPU box): scripts/trace_run.sh <tag> [bench args]  -> gpurun_out/<tag>/<tag>_kernel_trace.csv (+ stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
export TMPDIR=/tmp; cd /tmp; mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o $tag -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$tag/bench.json 2> $R/gpurun_out/$tag/err.log
echo "rc=$?"
