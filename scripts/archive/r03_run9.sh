#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_j; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
B="python bench.py --steps 5 --warmup 1 --distinct 128 --no-cpu-baseline"
timeout 400 $B > $o/bench.json 2> $o/bench.err; python - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"))
print(r["per_kernel_exclusive_us_per_image"], round(sum(r["per_kernel_exclusive_us_per_image"].values()),2))
print({k:v["ms_per_image"] for k,v in r["per_kernel_in_timed_region"].items()})
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $o/e2e -o e2e -- python $R/bench.py --no-extra-legs --no-cpu-baseline --distinct 128 --steps 3 --warmup 1 > $o/e2e.json 2> $o/e2e.err; cd $R
python profiles/summarize_csv.py stats $o/e2e | head -10
