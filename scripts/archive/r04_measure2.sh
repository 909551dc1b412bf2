#!/bin/bash
# Round-4 measurement set 2: GPU suite incl. the arithmetic-coded sources; WRITE's slot layout A/B (dword rows against 16-byte chunks: exclusive
# kernel times + SQ LDS counters); the one-image route with the larger engine pool; the format workloads and CPU baselines inside the
# container's CPU quota.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m2}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
run() { tag=$1; shift; timeout 900 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 3000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
for v in base rows base rows; do
  n=$((n+1))
  if [ $v = rows ]; then export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_rows.so; else unset LILLIPUT_HIP_LIB; fi
  run ab_${v}_$n python bench.py --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
done
unset LILLIPUT_HIP_LIB
LILLIPUT_HIP_COALESCE=0 run abi_direct python bench.py --workload abi --threads 8,32,64 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run abi_coalesced python bench.py --workload abi --threads 16,64,256 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run png2webp python bench.py --workload png2webp --threads 16 --batch 1024 --steps 2 --warmup 1
run png2webp_t32 python bench.py --workload png2webp --threads 32 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run animated python bench.py --workload animated --threads 16 --batch 256 --steps 2 --warmup 1
run firehose python bench.py --workload firehose --steps 2 --warmup 1
run bench_default python bench.py --steps 5 --warmup 1
# SQ counters of the two slot layouts (one stream, counters serialise the dispatches)
cd /tmp
for v in base rows; do
  if [ $v = rows ]; then export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_rows.so; else unset LILLIPUT_HIP_LIB; fi
  B="python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 256"
  LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/sq_$v -o sq -- $B > $o/sq_$v.json 2> $o/sq_$v.err || echo "sq $v failed: $(tail -2 $o/sq_$v.err)"
done
unset LILLIPUT_HIP_LIB
cd $R
python - <<PY
import json, glob, os, csv
from collections import defaultdict
for f in sorted(glob.glob("$o/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(os.path.basename(f), d["value"], d["unit"], "cpu", (d.get("cpu_baseline") or {}).get("value"), "frac", r.get("frac"), d["config"].get("verified_identical"),
              d["config"].get("resident_images_per_s"), r.get("per_kernel_exclusive_us_per_image"), d["config"].get("by_threads") and {k: v["images_per_s"] for k, v in d["config"]["by_threads"].items()})
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
for v in ("base", "rows"):
    hits = glob.glob("$o/sq_%s/**/*counter_collection.csv" % v, recursive=True)
    if not hits: continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(hits[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:40], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    names = sorted({k[0] for k in acc if k[0].startswith("k_huff")})
    for n in names:
        print(v, n, {c: round(acc[(n, c)] / max(1, cnt[(n, c)])) for (nn, c) in acc if nn == n})
    for h in hits: os.remove(h)
PY
