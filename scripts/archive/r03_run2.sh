#!/bin/bash
# round 3, GPU call 2: GPU test suite, default bench (sources in a pinned arena, copies spread over four queues) against the pageable route
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_c; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $o/pytest.log; cat $o/pytest.log
timeout 900 python bench.py --steps 10 --warmup 2 > $o/bench_pinned.json 2> $o/bench_pinned.err; echo "bench pinned rc=$?"; tail -c 300 $o/bench_pinned.err
timeout 300 python bench.py --steps 10 --warmup 2 --ingest pageable --no-extra-legs --no-cpu-baseline > $o/bench_pageable.json 2> $o/bench_pageable.err; echo "bench pageable rc=$?"
LILLIPUT_HIP_DIRECT_COPY_QUEUES=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline > $o/bench_pinned_q1.json 2> $o/bench_pinned_q1.err; echo "bench pinned, one copy queue rc=$?"
LILLIPUT_HIP_DIRECT_COPY_QUEUES=2 timeout 300 python bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline > $o/bench_pinned_q2.json 2> $o/bench_pinned_q2.err; echo "bench pinned, two copy queues rc=$?"
for f in pinned pageable pinned_q1 pinned_q2; do python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["config"].get("ingest"), "verified", d["config"].get("verified_outputs"), d["config"].get("verified_identical"), "h2d", d["config"].get("h2d_GBps_per_rank"), "resident", d["config"].get("resident_images_per_s"))
except Exception as e: print("$f", "no json", e)
PY
done
