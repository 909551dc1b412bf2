#!/bin/bash
# Round 6: resident rate of the headline sources against the engines per GPU and the images per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_resident; mkdir -p $O
for st in 2 4 6 8; do for ch in 0 28 56 85 170; do
    tag=${st}_${ch}
    LILLIPUT_HIP_STREAMS=$st timeout 300 python bench.py --resident --distinct 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --chunk $ch > $O/b_$tag.json 2> $O/b_$tag.err || tail -3 $O/b_$tag.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1])
    print("engines $st chunk %3s: resident %9.1f img/s  ms/step %.2f ok %s" % ("$ch", d["value"], d["ms_per_step"], d["config"].get("verified_identical")))
except Exception as e: print("$tag unreadable", e)
PY
done; done
