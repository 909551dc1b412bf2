#!/bin/bash
# Round 6: the integer scales around 1280^2 (5 x 5 boxes) on the closing library -- VERDICT r05 asked for a re-measurement -- and the q75 headline variant
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_sizes; mkdir -p $O
run() { tag=$1; shift
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --batch 2048 "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err || tail -5 $O/bench_$tag.err
    python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; c=d["config"]
pk=r["per_kernel_exclusive_us_per_image"]
print("%-8s e2e %9.1f resident %9s verified %s | %s | sum %.2f" % ("$tag", d["value"], c.get("resident_images_per_s"), c.get("verified_identical"), " ".join("%s %.2f" % (k.replace("k_",""), v) for k, v in pk.items()), sum(pk.values())))
PY
}
for s in 768 1024 1280 1536 1792 2048; do run $s --size $s; done
run 4096q75 --size 4096 --batch 1024 --source-quality 75
run 4000 --size 4000 --batch 1024
