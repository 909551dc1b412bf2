#!/bin/bash
# usage (GPU box): scripts/trace_cmd.sh <tag> <command...>  -> gpurun_out/<tag>/<tag>_kernel_stats.csv for an arbitrary command
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
export TMPDIR=/tmp; cd /tmp; mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o $tag -- "$@" > $R/gpurun_out/$tag/out.log 2> $R/gpurun_out/$tag/err.log
echo "rc=$?"
