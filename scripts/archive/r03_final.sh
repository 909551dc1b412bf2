#!/bin/bash
# round-3 measurement set (GPU box): default bench line, exclusive kernel statistics + SQ / TCC counters (scripts/pmc_sq.sh), kernel
# statistics of the pipelined run, the firehose workload, the host-side ingest rehearsal
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-r03_fin}; o=$R/gpurun_out/$tag; mkdir -p $o
export TMPDIR=/tmp; cd $R
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"; tail -c 300 $o/bench_default.err
timeout 300 python bench.py --ingest pageable --no-extra-legs --no-cpu-baseline > $o/bench_pageable.json 2> $o/bench_pageable.err; echo "pageable rc=$?"
bash scripts/pmc_sq.sh ${tag}_sq --batch 226 > $o/pmc_sq.log 2>&1; tail -3 $o/pmc_sq.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/e2e -o e2e -- python $R/bench.py --no-extra-legs --no-cpu-baseline --distinct 128 --steps 3 --warmup 1 > $o/e2e.json 2> $o/e2e.err
rm -f $o/*/*kernel_trace.csv $R/gpurun_out/${tag}_sq/*/*kernel_trace.csv $R/gpurun_out/${tag}_sq/*/*/*kernel_trace.csv
cd $R
timeout 600 python bench.py --workload firehose --steps 3 --warmup 1 --distinct 192 > $o/firehose.json 2> $o/firehose.err; echo "firehose rc=$?"
LD_LIBRARY_PATH=$R/lilliput_amd timeout 200 $R/scripts/ingest_scale /tmp/lilliput_bench_4096_q90 2.0 > $o/ingest_scale.md 2> $o/ingest_scale.err; cat $o/ingest_scale.md
python - <<PY
import json
for f in ("bench_default","bench_pageable","e2e","firehose"):
    try:
        d=json.loads(open("$o/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["config"].get("verified_identical"), (d.get("roofline") or {}).get("frac"), d["config"].get("resident_images_per_s"), d["config"].get("ingest",{}).get("mode") if isinstance(d["config"].get("ingest"),dict) else "")
    except Exception as e: print(f, "no json", e)
PY
