#!/bin/bash
# the reduced-instruction decode step: parity suite, then the default bench (end to end, resident, exclusive kernel times)
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_l}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $o/pytest.log; cat $o/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"))
print("roofline", d["roofline"])
for k,v in d["config"].items():
    if "kernel" in k or "us_per" in k: print(k, v)
PY
tail -5 $o/bench.err
