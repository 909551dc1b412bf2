#!/bin/bash
# usage (GPU box): scripts/r06_closing_lines.sh -> gpurun_out/r06_closing/*.json : the kept bench lines of the closing library
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r06_closing; mkdir -p $o; cd $R
run() { name=$1; shift; timeout 1200 "$@" > $o/$name.json 2> $o/$name.err; echo "$name rc=$? $(tail -c 200 $o/$name.err | tr '\n' ' ' | cut -c1-160)"; }
run headline python bench.py
run resident python bench.py --resident --no-cpu-baseline
run q75 python bench.py --source-quality 75 --no-cpu-baseline --distinct 128
run prog1024 python bench.py --source-sampling 420p --size 1024 --batch 256 --distinct 256 --steps 8 --warmup 2 --no-extra-legs
run prog4096 python bench.py --source-sampling 420p --size 4096 --batch 256 --distinct 64 --steps 3 --warmup 1 --no-extra-legs
run dri1 python bench.py --distinct 128 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --restart-rows 1
run size4000 python bench.py --size 4000 --distinct 128 --no-cpu-baseline
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06_closing/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], "resident", d["config"].get("resident_images_per_s"), "frac", d["roofline"].get("frac"), "traffic", d["roofline"].get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "ok", d["config"].get("verified_identical"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
