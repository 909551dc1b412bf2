#!/bin/bash
# r05: k_png_unfilter (byte-granular diagonal per channel, mailbox hand-over between bands) -- PNG parity tests, configs[2] at 16 callers, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/png_ab; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_png.py tests/test_png_output.py tests/test_firehose.py tests/test_handover.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 300 python bench.py --workload png2webp --threads 16 --steps 3 --warmup 1 > $O/bench_main.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload png2webp --threads 4 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -12 "$f" > $O/stats.csv
rm -rf $O/prof
tail -4 $O/pytest.log; tail -1 $O/bench_main.log | cut -c1-300; cut -d, -f1-6 $O/stats.csv
