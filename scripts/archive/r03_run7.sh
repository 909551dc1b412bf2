#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_h; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $o/pytest.log; cat $o/pytest.log
LILLIPUT_HIP_TRACE=1 timeout 600 python bench.py --workload firehose --steps 3 --warmup 1 --distinct 192 > $o/firehose.json 2> $o/firehose.err; echo "firehose rc=$?"; grep -v "chunk \|part:" $o/firehose.err | tail -8
python - <<PY
import json
d=json.loads(open("$o/firehose.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["items_per_format"], d["config"]["ok_per_format"], d["config"]["verified_outputs_per_format"], d["config"]["verified_identical"], d.get("cpu_baseline"))
PY
