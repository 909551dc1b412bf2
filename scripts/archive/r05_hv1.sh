#!/bin/bash
# r05: k_resample_hv1 (4:4:4 / 4:2:2 sources at integer scales) before / after the packed arithmetic: parity tests + exclusive kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/hv1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_area_fused.py tests/test_gpu_sweep.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for ss in 444 422; do
  for v in new old; do
    if [ $v = new ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$PWD/lilliput_amd/liblilliput_hip_hv1old.so; fi
    timeout 600 python bench.py --source-sampling $ss --distinct 64 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${ss}_$v.log 2> $O/bench_${ss}_$v.err
  done
done
unset LILLIPUT_HIP_LIB
tail -3 $O/pytest.log
python3 - <<'PY'
import json
for ss in ("444","422"):
    for v in ("new","old"):
        try:
            d=json.loads(open(f"gpurun_out/hv1/bench_{ss}_{v}.log").read().strip().splitlines()[-1])
            print(ss, v, d["value"], d["config"].get("resident_images_per_s"), d["config"].get("verified_identical"), d["roofline"].get("per_kernel_exclusive_us_per_image"))
        except Exception as e: print(ss, v, "failed", e)
PY
