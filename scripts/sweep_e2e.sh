#!/bin/bash
# usage (GPU box): scripts/sweep_e2e.sh <outdir> "<env assignments> -- <bench args>" ...   end-to-end bench variants, 64 distinct sources
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/$1; shift; mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1)); envs=${v%%--*}; args=${v#*--}
  env $envs python $R/bench.py --steps 3 --warmup 1 --distinct 64 --no-cpu-baseline --no-extra-legs $args > $out/v$i.json 2> $out/v$i.err
  python - "$v" $out/v$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); c=d["config"]
    print(sys.argv[1].ljust(50), d["value"], d["ms_per_step"], c.get("h2d_GBps_per_rank"), c.get("verify_rounds"), {k.split(" ")[0]:round(x["ms_per_image"]*1000,1) for k,x in d["roofline"]["per_kernel_in_timed_region"].items()})
except Exception as e: print(sys.argv[1], "failed", e)
PY
done
