#!/bin/bash
# chunks that taper towards the END of a call (16 / 8 / 4 images per engine), A/B on the headline workload
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_x}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 600 python -m pytest tests/test_ingest.py tests/test_firehose.py -m gpu -q -x 2>&1 | tail -4 ) > $o/pytest.log; cat $o/pytest.log
B="python bench.py --steps 10 --warmup 2 --distinct 256 --no-extra-legs --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B > $o/bench_$tag.json 2> $o/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", d["value"], d["ms_per_step"], "verified", d["config"].get("verified_identical"), "h2d", d["config"].get("h2d_GBps_per_rank"))
except Exception as e: print("$tag", "no json", e); print(open("$o/bench_$tag.err").read()[-600:])
PY
}
run taper A=1
run flat LILLIPUT_HIP_PIPE_TAPER=0
run taper_b A=1
run flat_b LILLIPUT_HIP_PIPE_TAPER=0
LILLIPUT_HIP_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --distinct 256 --no-extra-legs --no-cpu-baseline > $o/trace_taper.json 2> $o/trace_taper.err
grep "chunk \|run of" $o/trace_taper.err | tail -16
