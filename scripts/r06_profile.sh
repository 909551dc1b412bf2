#!/bin/bash
# Round-6 profile run on one MI355X: exclusive kernel statistics of the headline workload (rocprofv3 --kernel-trace --stats, one stream,
# launches of 113 images = one full round of WRITE workgroups), the SQ / TCC counter passes each in their own run (scripts/pmc_sq.sh), the summaries, then the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-r06_sq}
bash $R/scripts/pmc_sq.sh $tag --batch 226 > $R/gpurun_out/${tag}_run.log 2>&1
cd $R
python scripts/pmc_traffic.py gpurun_out/$tag 113 gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_traffic.md 2>&1
python profiles/summarize_csv.py stats gpurun_out/$tag/trace > gpurun_out/${tag}_kernel_stats.md 2>&1 || true
python profiles/summarize_sq.py gpurun_out/$tag 113 gpurun_out/${tag}_sq.json > gpurun_out/${tag}_sq.md 2>&1 || true
head -24 gpurun_out/${tag}_kernel_stats.md; head -16 gpurun_out/${tag}_traffic.md
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/${tag}_bench.err
