#!/bin/bash
# round-end measurement set (GPU box): default bench line, then kernel statistics of the end-to-end and the resident 4-engine runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-r02_e}; o=$R/gpurun_out/$tag; mkdir -p $o
export TMPDIR=/tmp; cd $R
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"; tail -c 400 $o/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/e2e -o e2e -- python $R/bench.py --no-extra-legs --no-cpu-baseline --distinct 64 --steps 3 --warmup 1 > $o/e2e.json 2> $o/e2e.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/res -o res -- python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 3 --warmup 1 > $o/res.json 2> $o/res.err
rm -f $o/*/*kernel_trace.csv
ls $o $o/e2e $o/res
