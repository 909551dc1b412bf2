#!/bin/bash
# Round 5: kernel statistics for SMALL sources (512 x 512 -> 256 x 256), one stream, resident: where the 6 us per image go
R=${GRAFT_REPO_ROOT:-$(pwd)}; sz=${1:-512}; o=$R/gpurun_out/r05_small_$sz; mkdir -p $o
export TMPDIR=/tmp LILLIPUT_HIP_STREAMS=1; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o trace -- python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 1024 --size $sz > $o/trace.json 2> $o/trace.err
cd $R; python profiles/summarize_csv.py stats $o/trace > $o/kernel_stats.md 2>&1; head -32 $o/kernel_stats.md
rm -f $o/trace/*kernel_trace.csv
