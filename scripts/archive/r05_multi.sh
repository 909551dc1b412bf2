#!/bin/bash
# Round 5: multi-symbol counting passes (LP_MULTI) against the one-symbol build: decode parity tests on the new library, then the
# exclusive per-kernel table of both builds on the headline sources (q90) and on sources of a photograph's density (q75).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r05_multi
O=gpurun_out/r05_multi
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_damaged.py tests/test_ingest.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in "$@"; do
  for q in 90 75; do
    if [ $v = default ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
    timeout 400 python bench.py --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --source-quality $q > $O/bench_${v}_q$q.json 2> $O/bench_${v}_q$q.err || tail -5 $O/bench_${v}_q$q.err
    python - <<PY
import json
d=json.loads(open("$O/bench_${v}_q$q.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$v q$q", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "launch", r["launch_images"], r["avg_launch_us"], "frac", r["frac"])
pk=r["per_kernel_exclusive_us_per_image"]; print(pk, "sum", round(sum(pk.values()),2))
PY
  done
done
unset LILLIPUT_HIP_LIB
