#!/bin/bash
# quick A/B: bench.py on 128 distinct sources, print the headline numbers and the exclusive per-kernel table
# usage (through gpurun): bash scripts/bench_quick.sh <tag> [extra bench args]
tag=$1; shift
mkdir -p gpurun_out/r02_d
timeout 500 python bench.py --distinct 128 --steps 2 --warmup 1 "$@" > gpurun_out/r02_d/bench_$tag.json 2> gpurun_out/r02_d/bench_$tag.err || tail -5 gpurun_out/r02_d/bench_$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_d/bench_$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$tag", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "launch", r["launch_images"], r["avg_launch_us"], "frac", r["frac"])
pk=r["per_kernel_exclusive_us_per_image"]; print(pk, "sum", round(sum(pk.values()),2))
PY
