#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/pmc_run.sh <tag> "<counter list>" [bench args...]
# One PMC pass (counters in their own run, kernel-trace only) over a short bench.py run; CSV output under gpurun_out/<tag>/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
ctrs=$1; shift
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/$tag
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/$tag -o $tag -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$tag/bench.json 2> $R/gpurun_out/$tag/err.log
echo "rc=$?"
