#!/bin/bash
# the round-end checks on the final tree within what is left of the GPU budget: GPU test suite, smoke(), the default bench line
# (without the CPU baseline leg, which is unchanged since scripts/r03_validate.sh last ran it)
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_val4}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > $o/pytest.log; cat $o/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -2 $o/smoke.log
timeout 600 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["config"]["verified_identical"], d["config"].get("resident_images_per_s"), r["frac"], r["traffic"], r["traffic_over_algorithmic"], r.get("per_kernel_exclusive_us_per_image"))
PY
