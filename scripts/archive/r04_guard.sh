#!/bin/bash
# the GPU test suite, smoke() and every bench mode with every device / pinned buffer of the library ending flush
# against an unmapped page (LILLIPUT_HIP_GUARD, lilliput_amd/csrc/lp_guard.h). A kernel or DMA transfer that touches one byte past a
# buffer aborts the process with "Memory access fault ... on address X"; with the allocation log (LILLIPUT_HIP_GUARD_LOG=1) X names
# the buffer. One pytest process per test file so that a fault costs one file, not the suite.
#   usage (GPU box, via gpurun): scripts/r04_guard.sh <tag> [alignment]
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_guard}; mkdir -p $o
A=${2:-64}
export TMPDIR=/tmp; cd $R
export LILLIPUT_HIP_GUARD=$A LILLIPUT_HIP_GUARD_LOG=1
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx" > $o/box.txt; nproc >> $o/box.txt; numactl -H 2>/dev/null | head -12 >> $o/box.txt
echo "== device ASan probe (HSA_XNACK=1)" | tee $o/summary.txt
if [ -z "$NOASAN" ]; then
( HSA_XNACK=1 LD_LIBRARY_PATH=$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)):$LD_LIBRARY_PATH timeout 60 ./scripts/asan_probe 2>&1 | tail -8 ) | tee -a $o/summary.txt
fi
echo "== smoke" | tee -a $o/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.out 2> $o/smoke.err; echo "smoke rc=$?" | tee -a $o/summary.txt
grep -E "fault|CANARY|smoke ok" $o/smoke.out $o/smoke.err | tail -5 | tee -a $o/summary.txt
for f in tests/test_*.py; do
  n=$(basename $f .py)
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $n "; then continue; fi
  timeout 900 python -m pytest $f -q -m gpu -p no:cacheprovider -v -s > $o/$n.out 2> $o/$n.err; rc=$?   # -s: the runtime's fault line and the guard's log are not captured away
  echo "$n rc=$rc $(grep -aE "passed|failed|deselected" $o/$n.out | tail -1)" | tee -a $o/summary.txt
  grep -ahE "Memory access fault|CANARY|did not take" $o/$n.out $o/$n.err | head -8 | tee -a $o/summary.txt
  if [ $rc -ne 0 ] && [ $rc -ne 5 ]; then tail -c 400000 $o/$n.err > $o/$n.err.tail; grep -av "guard\] \(dev\|host\)" $o/$n.out | tail -c 200000 > $o/$n.out.tail; fi
  rm -f $o/$n.err   # the allocation log is large; the tail is kept for a file that failed
  grep -aE "PASSED|FAILED|ERROR|SKIPPED" $o/$n.out | tail -400 > $o/$n.tests; rm -f $o/$n.out
done
echo "== bench modes under the guard (small batches: guarded buffers are allocated at exact size, so every chunk re-allocates)" | tee -a $o/summary.txt
if [ -n "$NOBENCH" ]; then exit 0; fi
for mode in "" "--size 4000" "--orientation 6" "--ingest pageable" "--workload firehose --batch 128" "--workload abi --threads 16 --batch 64" "--workload png2webp --threads 8 --batch 64" "--workload animated --threads 8 --batch 16"; do
  tag=$(echo "default $mode" | tr -c 'a-zA-Z0-9\n' '_')
  timeout 900 python bench.py --batch 128 --distinct 64 --steps 1 --warmup 1 --no-cpu-baseline $mode > $o/bench_$tag.json 2> $o/bench_$tag.err; rc=$?
  echo "bench [$mode] rc=$rc $(python -c "import json,sys; d=json.loads(open('$o/bench_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['config'].get('verified_identical'))" 2>/dev/null)" | tee -a $o/summary.txt
  grep -hE "Memory access fault|CANARY|GATE" $o/bench_$tag.err | head -5 | tee -a $o/summary.txt
  tail -c 200000 $o/bench_$tag.err > $o/bench_$tag.err.tail; rm -f $o/bench_$tag.err
done
python - <<PY | tee -a $o/summary.txt
import ctypes, sys
sys.path.insert(0, "$R")
import lilliput_amd as la
st = (ctypes.c_size_t * 4)()
la.lib().lilliput_hip_guard_stats(st)
print("guard stats of a fresh process (alignment, allocations, canary violations, peak mapped bytes):", list(st))
PY
