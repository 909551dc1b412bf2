#!/bin/bash
# Round-4 measurement set 3: who owns WRITE's LDS bank conflicts (timing-experiment builds: no coefficient stores / no flush; SQ counters), the
# service path with sleeping instead of spinning waits, format workloads by caller count, the firehose at BASELINE configs[4]'s stated scale.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m3}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 3000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
for t in 1 4 16; do
  run png2webp_t$t python bench.py --workload png2webp --threads $t --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
  LILLIPUT_HIP_BLOCKING_SYNC=1 run png2webp_block_t$t python bench.py --workload png2webp --threads $t --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
done
LILLIPUT_HIP_BLOCKING_SYNC=1 run animated_block_t16 python bench.py --workload animated --threads 16 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_BLOCKING_SYNC=1 run abi_coalesced_block python bench.py --workload abi --threads 16,64,256 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_BLOCKING_SYNC=1 run bench_default_block python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_BLOCKING_SYNC=1 run firehose_block python bench.py --workload firehose --steps 2 --warmup 1 --no-cpu-baseline
run firehose_scale python bench.py --workload firehose --batch 100000 --window 4096 --max-side 8192 --distinct 256 --steps 1 --warmup 0
cd /tmp
for v in noput noflush; do
  export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so
  B="python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 256 --verify 0"
  LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/sq_$v -o sq -- $B > $o/sq_$v.json 2> $o/sq_$v.err || echo "sq $v failed: $(tail -2 $o/sq_$v.err)"
  LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/tr_$v -o tr -- $B > $o/tr_$v.json 2> $o/tr_$v.err
done
unset LILLIPUT_HIP_LIB
cd $R
for v in noput noflush; do python profiles/summarize_csv.py stats $o/tr_$v 2>/dev/null | grep -E "huff|kernel" > $o/tr_$v.md; find $o/tr_$v -name "*.csv" -delete; done
python - <<PY
import json, glob, os, csv
from collections import defaultdict
for f in sorted(glob.glob("$o/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]
        print(os.path.basename(f), d["value"], d["unit"], "cpu", (d.get("cpu_baseline") or {}).get("value"), c.get("verified_identical"), "p50", c.get("request_latency_ms_p50"), "p99", c.get("request_latency_ms_p99"),
              c.get("by_threads") and {k: (v["images_per_s"], v["latency_ms_p50"]) for k, v in c["by_threads"].items()}, c.get("items_per_s_per_format"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
for v in ("noput", "noflush"):
    hits = glob.glob("$o/sq_%s/**/*counter_collection.csv" % v, recursive=True)
    if not hits: continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(hits[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:40], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for n in sorted({k[0] for k in acc if k[0].startswith("k_huff_write")}):
        print(v, n, {c: round(acc[(n, c)] / max(1, cnt[(n, c)])) for (nn, c) in acc if nn == n})
    for h in hits: os.remove(h)
    print(open("$o/tr_%s.md" % v).read())
PY
