#!/bin/bash
# fused fractional-area route: its GPU tests, then the 4000x4000 workload with the route on and off
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_area}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 300 python -m pytest tests/test_area_fused.py -x -q -m gpu 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
timeout 300 python bench.py --size 4000 --distinct 64 --batch 512 --no-cpu-baseline > $o/bench_on.json 2> $o/bench_on.err; echo "on rc=$?"
LILLIPUT_HIP_AREA_FUSED=0 timeout 300 python bench.py --size 4000 --distinct 64 --batch 512 --no-cpu-baseline > $o/bench_off.json 2> $o/bench_off.err; echo "off rc=$?"
python - <<PY
import json
for t in ("on","off"):
    try:
        d=json.loads(open("$o/bench_%s.json"%t).read().strip().splitlines()[-1]); r=d["roofline"]
        print(t, d["value"], d["ms_per_step"], d["config"]["verified_identical"], d["config"].get("resident_images_per_s"), r.get("per_kernel_exclusive_us_per_image"))
    except Exception as e: print(t, "failed", e)
PY
