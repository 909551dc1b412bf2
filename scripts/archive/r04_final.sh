#!/bin/bash
# Round-4 closing run on one fresh MI355X: GPU suite, smoke(), the suite and two bench modes under the guard allocator, the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_final}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $o/pytest.log; cat $o/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -3 $o/smoke.log
( LILLIPUT_HIP_GUARD=64 timeout 900 python -m pytest tests -q -m gpu -s -p no:cacheprovider 2>&1 | grep -aE "passed|failed|CANARY|Memory access fault|did not take" | tail -8 ) > $o/pytest_guard.log; echo "== guard"; cat $o/pytest_guard.log
for mode in "--size 4000"; do
  tag=$(echo "default $mode" | tr -c 'a-zA-Z0-9\n' '_')
  LILLIPUT_HIP_GUARD=64 timeout 600 python bench.py --batch 128 --distinct 64 --steps 1 --warmup 1 --no-cpu-baseline $mode > $o/guard_bench_$tag.json 2> $o/guard_bench_$tag.err; rc=$?
  echo "guard bench [$mode] rc=$rc $(python -c "import json,sys; d=json.loads(open('$o/guard_bench_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['config'].get('verified_identical'))" 2>/dev/null) $(grep -hcE 'Memory access fault|CANARY' $o/guard_bench_$tag.err)"
  tail -c 3000 $o/guard_bench_$tag.err > $o/guard_bench_$tag.err.tail; rm -f $o/guard_bench_$tag.err
done
timeout 1200 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"; tail -c 3000 $o/bench_default.err > $o/bench_default.err.tail; rm -f $o/bench_default.err
tail -1 $o/bench_default.json | cut -c1-1500
