#!/bin/bash
# tapered chunk sizes at both ends of a call: A/B on the headline workload, with the chunk timeline of one step each
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_k; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 600 python -m pytest tests/test_ingest.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
B="python bench.py --steps 10 --warmup 2 --distinct 256 --no-extra-legs --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B > $o/bench_$tag.json 2> $o/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$tag.json").read().strip().splitlines()[-1]); i=d["config"].get("ingest")
    print("$tag", d["value"], d["ms_per_step"], "stager_ms", i["stager_thread_ms_per_step"], "wait_ms", i["compute_threads_waiting_ms_per_step"], "verified", d["config"].get("verified_identical"), "h2d", d["config"].get("h2d_GBps_per_rank"))
except Exception as e: print("$tag", "no json", e); print(open("$o/bench_$tag.err").read()[-600:])
PY
}
run taper A=1
run flat LILLIPUT_HIP_PIPE_TAPER=0
run taper_b A=1
run flat_b LILLIPUT_HIP_PIPE_TAPER=0
run taper_pageable_cmp A=1
timeout 300 $B --ingest pageable > $o/bench_pageable.json 2> $o/bench_pageable.err; python -c "
import json; d=json.loads(open('$o/bench_pageable.json').read().strip().splitlines()[-1]); print('pageable taper', d['value'], d['config']['h2d_GBps_per_rank'])"
# timelines (one traced run each, 3 steps)
LILLIPUT_HIP_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --distinct 256 --no-extra-legs --no-cpu-baseline > $o/trace_taper.json 2> $o/trace_taper.err
LILLIPUT_HIP_TRACE=1 LILLIPUT_HIP_PIPE_TAPER=0 timeout 300 python bench.py --steps 3 --warmup 1 --distinct 256 --no-extra-legs --no-cpu-baseline > $o/trace_flat.json 2> $o/trace_flat.err
grep "chunk \|run of" $o/trace_taper.err | tail -44
grep "chunk \|run of" $o/trace_flat.err | tail -34
