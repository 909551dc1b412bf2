#!/bin/bash
# usage (GPU box): scripts/trace_e2e.sh <tag> [bench args]  -> kernel + memory-copy trace of the end-to-end bench (timeline analysis: scripts/timeline.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
export TMPDIR=/tmp; cd /tmp; mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/$tag -o $tag -- python $R/bench.py --no-cpu-baseline --no-extra-legs --distinct 64 --steps 2 --warmup 1 "$@" > $R/gpurun_out/$tag/bench.json 2> $R/gpurun_out/$tag/err.log
echo "rc=$?"; ls -la $R/gpurun_out/$tag | head
python $R/scripts/timeline.py $R/gpurun_out/$tag > $R/gpurun_out/$tag/timeline.txt 2>&1; rm -f $R/gpurun_out/$tag/*kernel_trace.csv   # the raw trace is large
tail -40 $R/gpurun_out/$tag/timeline.txt
