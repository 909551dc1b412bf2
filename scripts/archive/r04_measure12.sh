#!/bin/bash
# Round-4 measurement set 12: k_area_420 / k_area_420t with dot-product colour terms and paired clamp + conversion; A/B against the previous build.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m12}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $o/pytest.log; cat $o/pytest.log
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 2000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
n=0
for v in area03 new area03 new; do
  n=$((n+1))
  if [ $v = new ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
  run s4000_${v}_$n python bench.py --size 4000 --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
  run s4000o6_${v}_$n python bench.py --size 4000 --orientation 6 --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
done
unset LILLIPUT_HIP_LIB
run s3000_new python bench.py --size 3000 --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
cd /tmp
B="python $R/bench.py --size 4000 --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 232"
LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/tr_new -o tr -- $B > $o/tr_new.json 2> $o/tr_new.err
cd $R; python profiles/summarize_csv.py stats $o/tr_new 2>/dev/null > $o/tr_new.md; find $o/tr_new -name "*.csv" -delete
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$o/s*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]; r = d.get("roofline") or {}
        print(os.path.basename(f), d["value"], c.get("verified_identical"), "resident", c.get("resident_images_per_s"), r.get("per_kernel_exclusive_us_per_image"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
head -12 $o/tr_new.md
