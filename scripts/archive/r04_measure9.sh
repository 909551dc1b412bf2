#!/bin/bash
# Round-4 measurement set 9: short chunks at both ends of a batch call (LILLIPUT_HIP_PIPE_RAMP), A/B on the headline workload.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m9}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $o/pytest.log; cat $o/pytest.log
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 3000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
n=0
for v in 0 1 0 1 0 1; do
  n=$((n+1))
  LILLIPUT_HIP_PIPE_RAMP=$v run ramp${v}_$n python bench.py --distinct 256 --batch 1024 --steps 6 --warmup 1 --no-cpu-baseline --no-extra-legs
done
for v in 0 1 0 1; do
  n=$((n+1))
  LILLIPUT_HIP_PIPE_RAMP=$v run pageable_ramp${v}_$n python bench.py --ingest pageable --distinct 256 --batch 1024 --steps 6 --warmup 1 --no-cpu-baseline --no-extra-legs
done
LILLIPUT_HIP_TRACE=1 LILLIPUT_HIP_PIPE_RAMP=1 timeout 600 python bench.py --distinct 256 --batch 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/trace1.json 2> $o/trace1.err; grep "chunk" $o/trace1.err | tail -45 > $o/trace1_chunks.txt; rm -f $o/trace1.err
LILLIPUT_HIP_TRACE=1 LILLIPUT_HIP_PIPE_RAMP=0 timeout 600 python bench.py --distinct 256 --batch 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/trace0.json 2> $o/trace0.err; grep "chunk" $o/trace0.err | tail -34 > $o/trace0_chunks.txt; rm -f $o/trace0.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$o/*ramp*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("verified_identical"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
head -12 $o/trace1_chunks.txt; tail -8 $o/trace1_chunks.txt; echo; head -4 $o/trace0_chunks.txt; tail -4 $o/trace0_chunks.txt
