#!/bin/bash
# Round-4 measurement set 7: k_resample_420's inner loop rewritten (dot-product colour terms, SDWA adds, four pixels per sad): parity,
# A/B against round 3's loop (rs03), the same without SDWA (rsnosdwa) and with a 6-wave register budget (rsw6); HIP-event brackets
# with and without a stream drain before every timestamp against the rocprofv3 kernel trace of the same launches.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m7}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
for v in rsnosdwa rsw6; do
  ( LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ingest.py -q -m gpu -k "fused or orientation or headline or saturated or transform_matches" 2>&1 | tail -4 ) > $o/pytest_$v.log; echo "== $v"; cat $o/pytest_$v.log
done
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 3000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
n=0
for v in rs03 new rsnosdwa rsw6 rs03 new rsnosdwa rsw6; do
  n=$((n+1))
  if [ $v = new ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
  run ab_${v}_$n python bench.py --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
done
unset LILLIPUT_HIP_LIB
LILLIPUT_HIP_TIMING_SYNC=1 run ab_new_drain python bench.py --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
cd /tmp
for v in rs03 new; do
  if [ $v = new ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
  B="python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 256"
  LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $o/sq_$v -o sq -- $B > $o/sq_$v.json 2> $o/sq_$v.err || echo "sq $v failed: $(tail -2 $o/sq_$v.err)"
  LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/tr_$v -o tr -- $B > $o/tr_$v.json 2> $o/tr_$v.err
done
unset LILLIPUT_HIP_LIB
cd $R
for v in rs03 new; do python profiles/summarize_csv.py stats $o/tr_$v 2>/dev/null > $o/tr_$v.md; find $o/tr_$v -name "*.csv" -delete; done
python - <<PY
import json, glob, os, csv
from collections import defaultdict
for f in sorted(glob.glob("$o/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]; r = d.get("roofline") or {}
        print(os.path.basename(f), d["value"], d["unit"], c.get("verified_identical"), "resident", c.get("resident_images_per_s"), r.get("per_kernel_exclusive_us_per_image"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
for v in ("rs03", "new"):
    hits = glob.glob("$o/sq_%s/**/*counter_collection.csv" % v, recursive=True)
    if not hits: continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(hits[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:40], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for n in sorted({k[0] for k in acc if k[0].startswith("k_resample") or k[0].startswith("k_idct")}):
        print(v, n, {c: round(acc[(n, c)] / max(1, cnt[(n, c)])) for (nn, c) in acc if nn == n})
    for h in hits: os.remove(h)
    print("\n".join(l for l in open("$o/tr_%s.md" % v).read().splitlines() if "huff" in l or "idct" in l or "resample" in l or "unstuff" in l or "dc_" in l or "enc" in l))
PY
