#!/bin/bash
# what the driver runs at round end, on the final tree: GPU test suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_val}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > $o/pytest.log; cat $o/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -2 $o/smoke.log
timeout 900 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d["cpu_baseline"]
print(d["value"], d["ms_per_step"], d["config"]["verified_identical"], d["config"].get("resident_images_per_s"), r["frac"], r["traffic"], r["traffic_over_algorithmic"], c["value"])
PY
