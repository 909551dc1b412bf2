#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_f; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $o/pytest.log; cat $o/pytest.log
B="python bench.py --steps 5 --warmup 1 --distinct 128 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 400 $B > $o/bench_$tag.json 2> $o/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$tag", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"), "dominant", r.get("kernel","")[:30], r["frac"])
    pk=r["per_kernel_exclusive_us_per_image"]; print("   ", pk, "sum", round(sum(pk.values()),2))
except Exception as e: print("$tag", "no json", e); print(open("$o/bench_$tag.err").read()[-800:])
PY
}
run tok A=1
run write LILLIPUT_HIP_ENTROPY=write
