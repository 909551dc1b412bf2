#!/bin/bash
# usage (GPU box): scripts/r06_ab.sh [-q "90 75"] default v1 v2 ... : exclusive per-kernel us per image (bench.py's one-engine leg, HIP events) of the shipped
# library and of A/B builds lilliput_amd/liblilliput_hip_<v>.so (make -C lilliput_amd/csrc variant NAME=<v> DEFS=...); the bench's oracle gate
# (8 outputs byte-identical to the reference CPU path) runs for each. Results: gpurun_out/r06_ab/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_ab; mkdir -p $O
Q="90"; if [ "$1" = "-q" ]; then Q="$2"; shift 2; fi
for rep in 1 2; do
for v in "$@"; do
  for q in $Q; do
    if [ $v = default ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --source-quality $q > $O/bench_${v}_q${q}_$rep.json 2> $O/bench_${v}_q${q}_$rep.err || tail -5 $O/bench_${v}_q${q}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${v}_q${q}_$rep.json").read().strip().splitlines()[-1])
    r=d["roofline"]; pk=r["per_kernel_exclusive_us_per_image"]
    ent=sum(v for k,v in pk.items() if k.startswith(("k_huff","k_unstuff")))
    print("%-10s q$q rep$rep e2e %8.1f resident %s gate %s | write %.2f spec %.2f verify %.2f unstuff %.2f idct %.2f | entropy %.2f all %.2f" % ("$v", d["value"], d["config"].get("resident_images_per_s"), d["config"].get("verified_identical"),
          pk.get("k_huff_write",0), pk.get("k_huff_spec",0), pk.get("k_huff_verify",0), pk.get("k_unstuff_*",0), pk.get("k_idct",0), ent, sum(pk.values())))
except Exception as e:
    print("$v q$q rep$rep unreadable", e)
PY
  done
done
done
unset LILLIPUT_HIP_LIB
