#!/bin/bash
# Round-4 measurement set 11: the firehose's host side -- blocking waits and worker counts (environment only).
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m11}; mkdir -p $o
export TMPDIR=/tmp; cd $R
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 2000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
F="python bench.py --workload firehose --steps 3 --warmup 1 --no-cpu-baseline"
LILLIPUT_HIP_OTHER_WORKERS=32 LILLIPUT_HIP_BLOCKING_SYNC=1 run fh_w32_blocking $F
LILLIPUT_HIP_OTHER_WORKERS=48 LILLIPUT_HIP_BLOCKING_SYNC=1 run fh_w48_blocking $F
LILLIPUT_HIP_OTHER_WORKERS=24 LILLIPUT_HIP_BLOCKING_SYNC=1 run fh_w24_blocking $F
run fh_default $F
H="python bench.py --distinct 256 --batch 1024 --steps 6 --warmup 1 --no-cpu-baseline --no-extra-legs"
run head_default $H
LILLIPUT_HIP_BLOCKING_SYNC=1 run head_blocking $H
run head_pageable $H --ingest pageable
LILLIPUT_HIP_BLOCKING_SYNC=1 run head_pageable_blocking $H --ingest pageable
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$o/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(os.path.basename(f), d["value"], d["ms_per_step"], c.get("verified_identical"), c.get("items_per_s_per_format"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
