#!/bin/bash
# last GPU seconds of the round: smoke(), then as much of the GPU suite as fits (progress goes to the log unbuffered)
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_val5}; mkdir -p $o
export TMPDIR=/tmp PYTHONUNBUFFERED=1; cd $R
timeout 25 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -2 $o/smoke.log | cut -c1-300
timeout ${2:-26} python -m pytest tests/test_gpu_parity.py tests/test_area_fused.py tests/test_gpu_sweep.py tests -x -q -m gpu -p no:cacheprovider > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -c 600 $o/pytest.log
