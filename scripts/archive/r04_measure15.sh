#!/bin/bash
# Round-4 measurement set 15: paired-column colour path on top of the wide loads, A/B against wide loads alone.
# the suite under the guard allocator (the loads are not clamped into their rows), A/B against the dword loads.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m15}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $o/pytest.log; cat $o/pytest.log
( LILLIPUT_HIP_GUARD=64 timeout 900 python -m pytest tests/test_area_fused.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep -aE "passed|failed|CANARY|Memory access fault|did not take" | tail -8 ) > $o/pytest_guard.log; echo "== guard"; cat $o/pytest_guard.log
run() { tag=$1; shift; timeout 900 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 2000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
n=0
for v in areawide new areawide new; do
  n=$((n+1))
  if [ $v = new ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
  run s4000_${v}_$n python bench.py --size 4000 --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
  run s4000o6_${v}_$n python bench.py --size 4000 --orientation 6 --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
done
unset LILLIPUT_HIP_LIB
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$o/*s4000*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]; r = d.get("roofline") or {}
        print(os.path.basename(f), d["value"], c.get("verified_identical"), "resident", c.get("resident_images_per_s"), (r.get("per_kernel_exclusive_us_per_image") or {}).get("k_area_420"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
