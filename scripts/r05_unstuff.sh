#!/bin/bash
# Round 5: one-pass unstuff (ticket + decoupled look-back) against the three launches of round 1 (LILLIPUT_HIP_UNSTUFF=3pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r05_unstuff; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_damaged.py tests/test_ingest.py tests/test_progressive.py tests/test_multi_rank.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in fused 3pass fused 3pass; do
    if [ $v = fused ]; then unset LILLIPUT_HIP_UNSTUFF; else export LILLIPUT_HIP_UNSTUFF=3pass; fi
    for q in 90; do
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --source-quality $q > $O/bench_${v}_q$q.json 2> $O/bench_${v}_q$q.err || tail -5 $O/bench_${v}_q$q.err
    python - <<PY
import json
d=json.loads(open("$O/bench_${v}_q$q.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$v q$q", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"))
pk=r["per_kernel_exclusive_us_per_image"]; print(pk, "sum", round(sum(pk.values()),2))
PY
    done
done
unset LILLIPUT_HIP_UNSTUFF
