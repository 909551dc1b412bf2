#!/bin/bash
# Round-5 profile run on one MI355X: exclusive kernel statistics (rocprofv3 --kernel-trace --stats, one stream, launches of 113 images),
# the SQ / TCC counter passes each in their own run (scripts/pmc_sq.sh), then the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-r05_sq}
bash $R/scripts/pmc_sq.sh $tag --batch 226 > $R/gpurun_out/${tag}_run.log 2>&1
cd $R
python scripts/pmc_traffic.py gpurun_out/$tag 113 gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_traffic.md 2>&1
python profiles/summarize_csv.py stats gpurun_out/$tag/trace > gpurun_out/${tag}_kernel_stats.md 2>&1 || true
python profiles/summarize_sq.py gpurun_out/$tag 113 gpurun_out/${tag}_sq.json > gpurun_out/${tag}_sq.md 2>&1 || true
head -30 gpurun_out/${tag}_kernel_stats.md; head -20 gpurun_out/${tag}_traffic.md
