#!/bin/bash
# Round 6: subsequence size of the headline decode re-swept on the closing library (exclusive per-kernel us per image)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_subbits; mkdir -p $O
for rep in 1 2; do for sb in 0 4096 8192 16384 32768; do for cb in 0; do
    tag=${sb}_${cb}_$rep
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --sub-bits $sb --ckpt-bits $cb > $O/bench_$tag.json 2> $O/bench_$tag.err || tail -3 $O/bench_$tag.err
    python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; c=d["config"]; pk=r["per_kernel_exclusive_us_per_image"]
ent=sum(v for k,v in pk.items() if k.startswith(("k_huff","k_unstuff")))
print("S %6s C %4s rep $rep e2e %8.1f resident %9s ok %s | %s | entropy %.2f all %.2f launch %s" % ("$sb", "$cb", d["value"], c.get("resident_images_per_s"), c.get("verified_identical"), " ".join("%s %.2f" % (k.replace("k_",""), v) for k, v in pk.items() if "huff" in k or "unstuff" in k), ent, sum(pk.values()), r.get("launch_images")))
PY
done; done; done
