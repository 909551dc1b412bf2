#!/bin/bash
# queue-based verify: workgroups per image (share) and refill period
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_q}; mkdir -p $o
export TMPDIR=/tmp; cd $R
for v in s1 s2 s1r64 s2r16; do
  lib=$R/lilliput_amd/liblilliput_hip_$v.so
  LILLIPUT_HIP_LIB=$lib timeout 400 python bench.py --steps 6 --warmup 2 --distinct 256 --no-cpu-baseline > $o/bench_$v.json 2> $o/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"), d["roofline"]["per_kernel_exclusive_us_per_image"])
except Exception as e: print("$v", "no json", e); print(open("$o/bench_$v.err").read()[-500:])
PY
done
