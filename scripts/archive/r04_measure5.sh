#!/bin/bash
# Round-4 measurement set 5: where a PNG -> WebP request's time goes when 8 callers share the GPU (HIP API trace + kernel trace of
# bench.py --workload png2webp --threads 8), one pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m5}; mkdir -p $o
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d $o/api -o api -- python $R/bench.py --workload png2webp --threads 8 --batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/api.json 2> $o/api.err; echo "api rc=$?"
cd $R
for f in $(find $o/api -name "*_stats.csv"); do echo "== $(basename $f)"; head -25 $f; done > $o/api_stats.txt
find $o/api -name "*.csv" ! -name "*_stats.csv" -delete
cat $o/api_stats.txt | cut -c1-200
