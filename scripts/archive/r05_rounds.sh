#!/bin/bash
# Round 5: WRITE's throughput against its occupancy and against launches of more than one round of workgroups. Arguments: pad:chunk
# pairs -- LILLIPUT_HIP_WRITE_LDS_PAD (0 = four workgroups per CU, 4096 = three, 16384 = two) and LILLIPUT_HIP_RESIDENT_CHUNK (0 = one round).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_rounds
O=gpurun_out/r05_rounds
for pc in "$@"; do
    pad=${pc%%:*}; c=${pc##*:}
    export LILLIPUT_HIP_WRITE_LDS_PAD=$pad
    if [ $c = 0 ]; then unset LILLIPUT_HIP_RESIDENT_CHUNK; else export LILLIPUT_HIP_RESIDENT_CHUNK=$c; fi
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_p${pad}_c$c.json 2> $O/bench_p${pad}_c$c.err || tail -5 $O/bench_p${pad}_c$c.err
    python - <<PY
import json
d=json.loads(open("$O/bench_p${pad}_c$c.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("pad $pad chunk $c", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "launch", r["launch_images"], r["avg_launch_us"], "frac", r["frac"])
pk=r["per_kernel_exclusive_us_per_image"]; print(pk, "sum", round(sum(pk.values()),2))
PY
done
