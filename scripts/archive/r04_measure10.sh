#!/bin/bash
# Round-4 measurement set 10: dispatcher threads of the call coalescer (LILLIPUT_HIP_COALESCE_WORKERS = 2 / 3 / 4) at 16 / 64 / 256 callers.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m10}; mkdir -p $o
export TMPDIR=/tmp; cd $R
run() { tag=$1; shift; timeout 1500 "$@" > $o/$tag.json 2> $o/$tag.err; echo "$tag rc=$?"; tail -c 2000 $o/$tag.err > $o/$tag.err.tail; rm -f $o/$tag.err; }
for w in 4 2 3 4 2 3; do
  n=$((n+1))
  LILLIPUT_HIP_COALESCE_WORKERS=$w run abi_w${w}_$n python bench.py --workload abi --threads 64,256 --batch 1024 --distinct 256 --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$o/abi_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(os.path.basename(f), c.get("verified_identical"), {k: (v["images_per_s"], v["latency_ms_p50"], v.get("host_cpu_ms_per_request")) for k, v in c["by_threads"].items()})
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
