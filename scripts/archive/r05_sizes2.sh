#!/bin/bash
# Round 5: more shapes a service sees: other samplings at small integer scales, sources at / below the thumbnail size
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r05_sizes2; mkdir -p $O
run() { tag=$1; shift
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --batch 2048 "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err || tail -5 $O/bench_$tag.err
    python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; c=d["config"]
print("$tag", "e2e", d["value"], "resident", c.get("resident_images_per_s"), "verified", c.get("verified_identical"))
pk=r["per_kernel_exclusive_us_per_image"]; print(pk, "sum", round(sum(pk.values()),2))
PY
}
run 444_512 --size 512 --source-sampling 444
run 444_1024 --size 1024 --source-sampling 444
run 422_768 --size 768 --source-sampling 422
run 420_256 --size 256
run 420_128 --size 128
run 420_512_o6 --size 512 --orientation 6
run 420_1024_out128 --size 1024 --out 128
run 420_1024_out512 --size 1024 --out 512
