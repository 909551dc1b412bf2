#!/bin/bash
# Round-4 measurement set 1 (GPU box): the headline line with the C-harness CPU baseline, the drop-in path under concurrent callers
# (coalesced / direct), the configs[2] / [3] / [4] lines, the host-side scaling probe, and the rocprofv3 pass over the fused-area kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m1}; mkdir -p $o
export TMPDIR=/tmp; cd $R
run() { tag=$1; shift; timeout 900 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 4000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $o/pytest.log; cat $o/pytest.log
run bench_default python bench.py --steps 5 --warmup 1
run abi_coalesced python bench.py --workload abi --threads 1,2,4,8,16,64,256 --batch 1024 --steps 2 --warmup 1
LILLIPUT_HIP_COALESCE=0 run abi_direct python bench.py --workload abi --threads 1,8,64 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_COALESCE=0 LILLIPUT_HIP_ENGINE_POOL=64 run abi_direct_pool64 python bench.py --workload abi --threads 8,64 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_COALESCE=2 run abi_coalesce_from2 python bench.py --workload abi --threads 2,4,8 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run png2webp python bench.py --workload png2webp --threads 32 --batch 2048 --steps 2 --warmup 1
run animated python bench.py --workload animated --threads 32 --batch 256 --steps 2 --warmup 1
run firehose python bench.py --workload firehose --steps 2 --warmup 1
timeout 600 ./scripts/host_scale 1.0 > $o/host_scale.md 2>&1; echo "host_scale rc=$?"
cd /tmp
LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_area -o area -- python $R/bench.py --size 4000 --resident --batch 256 --distinct 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/prof_area.json 2> $o/prof_area.err; echo "prof_area rc=$?"
LILLIPUT_HIP_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_area6 -o area6 -- python $R/bench.py --size 4000 --orientation 6 --resident --batch 256 --distinct 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/prof_area6.json 2> $o/prof_area6.err; echo "prof_area6 rc=$?"
cd $R
for d in prof_area prof_area6; do python profiles/summarize_csv.py stats $o/$d > $o/$d.md 2>/dev/null; find $o/$d -name "*.csv" ! -name "*kernel_stats.csv" -delete; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$o/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["unit"], (d.get("cpu_baseline") or {}).get("value"), (d.get("roofline") or {}).get("frac"), d["config"].get("verified_identical"), d["config"].get("by_threads") and {k: v["images_per_s"] for k, v in d["config"]["by_threads"].items()})
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
