#!/bin/bash
# Round-4 measurement set 6: the one-image route after the transfer fixes (contiguous Mats as one copy, pageable host memory through the
# engines' pinned buffers): GPU suite, the same suite files under the guard, configs[2] / [3] by caller count, firehose, direct abi route.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m6}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
( LILLIPUT_HIP_GUARD=64 timeout 900 python -m pytest tests/test_png.py tests/test_webp.py tests/test_gif.py tests/test_thumbhash.py tests/test_png_output.py tests/test_gpu_parity.py tests/test_ingest.py -q -m gpu -s 2>&1 | grep -aE "passed|failed|CANARY|fault|did not take" | tail -8 ) > $o/pytest_guard.log; cat $o/pytest_guard.log
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 3000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
for t in 1 4 8 16; do run png2webp_t$t python bench.py --workload png2webp --threads $t --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs; done
run animated_t16 python bench.py --workload animated --threads 16 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run animated_t4 python bench.py --workload animated --threads 4 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
LILLIPUT_HIP_COALESCE=0 run abi_direct python bench.py --workload abi --threads 1,8,32 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
run firehose python bench.py --workload firehose --steps 2 --warmup 1 --no-cpu-baseline
run png2webp_full python bench.py --workload png2webp --threads 16 --batch 1024 --steps 2 --warmup 1
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$o/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(os.path.basename(f), d["value"], d["unit"], c.get("verified_identical"), "p50", c.get("request_latency_ms_p50"), "cpu ms/req", c.get("host_cpu_ms_per_request"), "cpus busy", c.get("host_cpus_busy"),
              "cpu_baseline", (d.get("cpu_baseline") or {}).get("value"), c.get("by_threads") and {k: (v["images_per_s"], v["latency_ms_p50"]) for k, v in c["by_threads"].items()})
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
