#!/bin/bash
# Round 5: source sizes other than the headline's (what a service sees): end to end and resident rate, exclusive per-kernel table.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r05_sizes; mkdir -p $O
for sz in "$@"; do
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --size $sz --batch 2048 > $O/bench_$sz.json 2> $O/bench_$sz.err || tail -5 $O/bench_$sz.err
    python - <<PY
import json
d=json.loads(open("$O/bench_$sz.json").read().strip().splitlines()[-1])
r=d["roofline"]; c=d["config"]
print("size $sz", "e2e", d["value"], "ms/step", d["ms_per_step"], "resident", c.get("resident_images_per_s"), "in bytes", c.get("mean_input_bytes"), "h2d", c.get("h2d_GBps_per_rank"), "launch", r["launch_images"], r["avg_launch_us"])
pk=r["per_kernel_exclusive_us_per_image"]; print(pk, "sum", round(sum(pk.values()),2))
print({k:v for k,v in c.get("ingest",{}).items() if k in ("stage_ms","stall_ms","wall_ms","threads")})
PY
done
