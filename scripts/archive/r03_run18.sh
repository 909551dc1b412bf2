#!/bin/bash
# wave lifetime against dispatch rate: tiles per wave of k_idct, chunks per workgroup of the unstuff kernels (variants built with make variant)
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_i}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_ingest.py -m gpu -q -x 2>&1 | tail -3 ) > $o/pytest.log; cat $o/pytest.log
for v in base u1 u8 u16; do
  lib=$R/lilliput_amd/liblilliput_hip_$v.so; [ $v = base ] && lib=$R/lilliput_amd/liblilliput_hip.so
  LILLIPUT_HIP_LIB=$lib timeout 400 python bench.py --steps 5 --warmup 2 --distinct 256 --no-cpu-baseline > $o/bench_$v.json 2> $o/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"), d["roofline"]["per_kernel_exclusive_us_per_image"])
except Exception as e: print("$v", "no json", e); print(open("$o/bench_$v.err").read()[-500:])
PY
done
