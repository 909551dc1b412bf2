#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_g; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $o/pytest.log; cat $o/pytest.log
B="python bench.py --resident --steps 2 --warmup 1 --distinct 64 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 400 $B > $o/bench_$tag.json 2> $o/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    pk=r["per_kernel_exclusive_us_per_image"]; print("$tag", "resident", d["value"], "verified", d["config"].get("verified_identical"), pk, "sum", round(sum(pk.values()),2))
except Exception as e: print("$tag", "no json", e); print(open("$o/bench_$tag.err").read()[-800:])
PY
grep "verify walks" $o/bench_$tag.err | tail -2
}
run tok A=1
run write LILLIPUT_HIP_ENTROPY=write
run notok LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_notok.so
run noflush LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_noflush.so
run dbg LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_dbg.so LILLIPUT_HIP_DEBUG_COUNTERS=1
