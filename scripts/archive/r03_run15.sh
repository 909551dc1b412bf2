#!/bin/bash
# subsequence size against lanes in flight, after the step rewrite (LILLIPUT_HIP_LAT_LANES = lanes a launch should have before S stops shrinking)
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_s}; mkdir -p $o
export TMPDIR=/tmp; cd $R
for v in 131072 262144 524288; do
  LILLIPUT_HIP_LAT_LANES=$v timeout 400 python bench.py --steps 6 --warmup 2 --distinct 256 --no-cpu-baseline > $o/bench_$v.json 2> $o/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"), d["roofline"]["frac"], d["roofline"]["per_kernel_exclusive_us_per_image"])
except Exception as e: print("$v", "no json", e); print(open("$o/bench_$v.err").read()[-500:])
PY
done
