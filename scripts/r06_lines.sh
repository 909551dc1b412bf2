#!/bin/bash
# usage (GPU box): scripts/r06_lines.sh  -> gpurun_out/r06_lines/*.json : the kept bench lines of round 6 that are not the headline
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r06_lines; mkdir -p $o; cd $R
run() { name=$1; shift; timeout 900 "$@" > $o/$name.json 2> $o/$name.err; echo "$name rc=$? $(tail -c 300 $o/$name.err | tr '\n' ' ' | cut -c1-200)"; }
run dri0 python bench.py --distinct 128 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs
run dri1 python bench.py --distinct 128 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --restart-rows 1
run firehose python bench.py --workload firehose --steps 2 --warmup 1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06_lines/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], json.dumps(d.get("roofline", {}))[:600])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
