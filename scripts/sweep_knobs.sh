#!/bin/bash
# usage (GPU box): scripts/sweep_knobs.sh <outdir> -- a few bench.py runs over the decoder's tuning knobs (subsequence size, chunk, streams)
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/$1; mkdir -p $out
run() { tag=$1; shift; env "$@" python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > $out/$tag.json 2>/dev/null; }
ARGS="--sub-bits 8192" run s1_S8192 LILLIPUT_HIP_STREAMS=1
ARGS="--sub-bits 4096" run s1_S4096 LILLIPUT_HIP_STREAMS=1
ARGS="--chunk 256" run s1_c256 LILLIPUT_HIP_STREAMS=1
ARGS="--chunk 256 --sub-bits 8192" run s1_c256_S8192 LILLIPUT_HIP_STREAMS=1
ARGS="--sub-bits 8192" run s4_S8192 LILLIPUT_HIP_STREAMS=4
ARGS="--chunk 64" run s4_c64 LILLIPUT_HIP_STREAMS=4
ARGS="--chunk 64" run s8_c64 LILLIPUT_HIP_STREAMS=8
ARGS="" run s8 LILLIPUT_HIP_STREAMS=8
ARGS="" run s2 LILLIPUT_HIP_STREAMS=2
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.load(open(f))
        print(os.path.basename(f)[:-5].ljust(16),d["value"],d["config"].get("verify_rounds"),{k.split(" ")[0]:round(x["ms_per_image"]*1000,2) for k,x in d["roofline"]["per_kernel"].items()})
    except Exception as e: print(f,"failed",e)
PY
