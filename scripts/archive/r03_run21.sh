#!/bin/bash
# fused fractional-area route, all orientations: its GPU tests, then the 4000x4000 workload upright and with orientation 6
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_area3}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 300 python -m pytest tests/test_area_fused.py -x -q -m gpu 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
timeout 300 python bench.py --size 4000 --distinct 64 --batch 512 --no-cpu-baseline > $o/bench_o1.json 2> $o/bench_o1.err; echo "o1 rc=$?"
timeout 300 python bench.py --size 4000 --distinct 64 --batch 512 --no-cpu-baseline --orientation 6 > $o/bench_o6.json 2> $o/bench_o6.err; echo "o6 rc=$?"
python - <<PY
import json
for t in ("o1","o6"):
    try:
        d=json.loads(open("$o/bench_%s.json"%t).read().strip().splitlines()[-1]); r=d["roofline"]
        print(t, d["value"], d["ms_per_step"], d["config"]["verified_identical"], d["config"].get("resident_images_per_s"), r.get("per_kernel_exclusive_us_per_image"))
    except Exception as e: print(t, "failed", e)
PY
