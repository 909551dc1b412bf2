#!/bin/bash
# The differential GPU tests on fresh seeds (tests/conftest.py LILLIPUT_FUZZ_RNG_OFFSET): scripts/r06_fresh_gpu.sh OFFSET...  (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $R; O=$R/gpurun_out/r06_fresh; mkdir -p $O
for k in "$@"; do
  LILLIPUT_FUZZ_RNG_OFFSET=$k timeout 900 python -m pytest tests/test_damaged.py tests/test_gpu_sweep.py tests/test_progressive.py tests/test_gpu_parity.py tests/test_arith.py tests/test_area_fused.py -q -m gpu -p no:cacheprovider > $O/offset_$k.log 2>&1
  echo "offset $k: $(tail -1 $O/offset_$k.log)"; grep "^FAILED" $O/offset_$k.log | cut -c1-200
done
