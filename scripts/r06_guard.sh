#!/bin/bash
# Round 6: the progressive device path (lp_kernels_prog.hip: clamped stream reads, block prefetch that re-reads the last block, merged
# per-file transfers with raw_skip) and the rest of the GPU suite with every device / pinned buffer ending flush against an unmapped page
# (LILLIPUT_HIP_GUARD, lp_guard.h). usage (GPU box): scripts/r06_guard.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r06_guard; mkdir -p $o; cd $R
export TMPDIR=/tmp LILLIPUT_HIP_GUARD=64
: > $o/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.out 2>&1; echo "smoke rc=$? $(grep -c 'smoke ok' $o/smoke.out)" | tee -a $o/summary.txt
for f in tests/test_*.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -q -m gpu -p no:cacheprovider > $o/$n.out 2>&1; rc=$?
  echo "$n rc=$rc $(grep -aE "passed|failed|deselected" $o/$n.out | tail -1)" | tee -a $o/summary.txt
  grep -ahE "Memory access fault|CANARY" $o/$n.out | head -4 | tee -a $o/summary.txt
done
for mode in "--source-sampling 420p --size 512 --batch 128 --distinct 32" "--source-sampling 420p --size 1024 --batch 96 --distinct 16 --restart-rows 1" "--size 1024 --batch 128 --distinct 32"; do
  tag=$(echo "$mode" | tr -c 'a-zA-Z0-9\n' '_')
  timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs $mode > $o/bench_$tag.json 2> $o/bench_$tag.err; rc=$?
  echo "bench [$mode] rc=$rc $(python -c "import json,sys; d=json.loads(open('$o/bench_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['config'].get('verified_identical'))" 2>/dev/null)" | tee -a $o/summary.txt
  grep -hE "Memory access fault|CANARY|GATE" $o/bench_$tag.err | head -3 | tee -a $o/summary.txt
done
