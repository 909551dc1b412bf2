#!/bin/bash
# usage (GPU box): scripts/r06_final_lines.sh -> gpurun_out/r06_final/*.json : the progressive-source lines of round 6 with their CPU baseline,
# after the PxM tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r06_final; mkdir -p $o; cd $R
timeout 900 python -m pytest tests/test_pxm.py tests/test_bmp.py -x -q -m gpu 2>&1 | tail -5
run() { name=$1; shift; timeout 900 "$@" > $o/$name.json 2> $o/$name.err; echo "$name rc=$? $(tail -c 300 $o/$name.err | tr '\n' ' ' | cut -c1-200)"; }
run prog1024 python bench.py --source-sampling 420p --size 1024 --batch 256 --distinct 256 --steps 8 --warmup 2 --no-extra-legs
run prog4096 python bench.py --source-sampling 420p --size 4096 --batch 256 --distinct 64 --steps 3 --warmup 1 --no-extra-legs
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06_final/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], json.dumps(d.get("cpu_baseline", {}))[:300])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
