#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_e; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 600 python -m pytest tests/test_ingest.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -x 2>&1 | tail -15 ) > $o/pytest.log; cat $o/pytest.log
B="python bench.py --steps 10 --warmup 2 --distinct 256 --no-extra-legs --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B > $o/bench_$tag.json 2> $o/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$tag.json").read().strip().splitlines()[-1]); i=d["config"].get("ingest")
    print("$tag", d["value"], i["mode"], "stager_ms", i["stager_thread_ms_per_step"], "wait_ms", i["compute_threads_waiting_ms_per_step"], "verified", d["config"].get("verified_identical"), "h2d", d["config"].get("h2d_GBps_per_rank"))
except Exception as e: print("$tag", "no json", e); print(open("$o/bench_$tag.err").read()[-600:])
PY
}
run merged A=1
run nomerge LILLIPUT_HIP_MERGE_GAP=0
run merged_c64 LILLIPUT_HIP_PIPE_CHUNK=64
run merged_c16 LILLIPUT_HIP_PIPE_CHUNK=16
run merged_s5 LILLIPUT_HIP_STREAMS=5
timeout 300 $B --ingest pageable > $o/bench_pageable.json 2> $o/bench_pageable.err; python -c "
import json; d=json.loads(open('$o/bench_pageable.json').read().strip().splitlines()[-1]); print('pageable', d['value'], d['config']['h2d_GBps_per_rank'])"
run merged_b A=1
