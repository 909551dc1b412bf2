#!/bin/bash
# usage (GPU box): scripts/sweep_pipe.sh <outdir> [variants...]  each variant "CHUNK:STREAMS"; end-to-end bench (host bytes in -> out), 64 distinct sources
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  c=${v%%:*}; s=${v##*:}
  LILLIPUT_HIP_PIPE_CHUNK=$c LILLIPUT_HIP_STREAMS=$s python $R/bench.py --steps 3 --warmup 1 --distinct 64 --no-cpu-baseline --no-extra-legs > $out/c${c}_s${s}.json 2> $out/c${c}_s${s}.err
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]
        print(os.path.basename(f)[:-5].ljust(12), d["value"], d["ms_per_step"], c.get("h2d_GBps_per_rank"), c.get("ingest"), {k.split(" ")[0]:round(x["ms_per_image"]*1000,1) for k,x in d["roofline"]["per_kernel_in_timed_region"].items()})
    except Exception as e: print(f,"failed",e)
PY
