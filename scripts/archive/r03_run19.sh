#!/bin/bash
# engines per GPU and images per chunk of the ingest pipeline, after the kernel work of round 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_p2}; mkdir -p $o
export TMPDIR=/tmp; cd $R
B="python bench.py --steps 10 --warmup 2 --distinct 256 --no-extra-legs --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B > $o/bench_$tag.json 2> $o/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", d["value"], d["ms_per_step"], "verified", d["config"].get("verified_identical"), "h2d", d["config"].get("h2d_GBps_per_rank"))
except Exception as e: print("$tag", "no json", e); print(open("$o/bench_$tag.err").read()[-600:])
PY
}
run s4 A=1
run s5 LILLIPUT_HIP_STREAMS=5
run s6 LILLIPUT_HIP_STREAMS=6
run s3 LILLIPUT_HIP_STREAMS=3
run c48 LILLIPUT_HIP_PIPE_CHUNK=48
run c24 LILLIPUT_HIP_PIPE_CHUNK=24
run s4b A=1
