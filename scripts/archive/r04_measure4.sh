#!/bin/bash
# Round-4 measurement set 4: padded slot stride in WRITE (A/B against 1024), the coalesced service path with callers staging their own
# sources, host CPU accounting of the service workloads.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m4}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 3000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
for v in s1024 pad s1024 pad; do
  n=$((n+1))
  if [ $v = s1024 ]; then export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_s1024.so; else unset LILLIPUT_HIP_LIB; fi
  run ab_${v}_$n python bench.py --distinct 128 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline
done
unset LILLIPUT_HIP_LIB
run abi_coalesced python bench.py --workload abi --threads 4,16,64,256 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_COALESCE_PINNED_MB=0 run abi_coalesced_nostage python bench.py --workload abi --threads 64,256 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_COALESCE_WORKERS=8 run abi_coalesced_w8 python bench.py --workload abi --threads 64,256 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run png2webp_t16 python bench.py --workload png2webp --threads 16 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run png2webp_t8 python bench.py --workload png2webp --threads 8 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run animated_t16 python bench.py --workload animated --threads 16 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
cd /tmp
for v in s1024 pad; do
  if [ $v = s1024 ]; then export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_s1024.so; else unset LILLIPUT_HIP_LIB; fi
  B="python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 256"
  LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/sq_$v -o sq -- $B > $o/sq_$v.json 2> $o/sq_$v.err || echo "sq $v failed: $(tail -2 $o/sq_$v.err)"
  LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/tr_$v -o tr -- $B > $o/tr_$v.json 2> $o/tr_$v.err
done
unset LILLIPUT_HIP_LIB
cd $R
for v in s1024 pad; do python profiles/summarize_csv.py stats $o/tr_$v 2>/dev/null > $o/tr_$v.md; find $o/tr_$v -name "*.csv" -delete; done
python - <<PY
import json, glob, os, csv
from collections import defaultdict
for f in sorted(glob.glob("$o/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]; r = d.get("roofline") or {}
        print(os.path.basename(f), d["value"], d["unit"], c.get("verified_identical"), "resident", c.get("resident_images_per_s"), r.get("per_kernel_exclusive_us_per_image"),
              "p50", c.get("request_latency_ms_p50"), "cpu ms/req", c.get("host_cpu_ms_per_request"), "cpus busy", c.get("host_cpus_busy"), "throttled", c.get("throttled_periods"),
              c.get("by_threads") and {k: (v["images_per_s"], v["latency_ms_p50"], v.get("host_cpu_ms_per_request"), v.get("host_cpus_busy"), v.get("throttled_periods")) for k, v in c["by_threads"].items()})
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
for v in ("s1024", "pad"):
    hits = glob.glob("$o/sq_%s/**/*counter_collection.csv" % v, recursive=True)
    if not hits: continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(hits[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:40], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for n in sorted({k[0] for k in acc if k[0].startswith("k_huff_write")}):
        print(v, n, {c: round(acc[(n, c)] / max(1, cnt[(n, c)])) for (nn, c) in acc if nn == n})
    for h in hits: os.remove(h)
    print("\n".join(l for l in open("$o/tr_%s.md" % v).read().splitlines() if "huff" in l or "idct" in l))
PY
