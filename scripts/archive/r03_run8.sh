#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_i; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests/test_png.py tests/test_png_output.py tests/test_firehose.py tests/test_color.py -m gpu -q -x 2>&1 | tail -8 ) > $o/pytest.log; cat $o/pytest.log
timeout 300 python scripts/png_bench.py 2>&1 | tail -4 | tee $o/png_bench.txt
LILLIPUT_HIP_PNG_ROWWISE=1 timeout 300 python scripts/png_bench.py 2>&1 | tail -4 | tee $o/png_bench_rowwise.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $o/pngprof -o png -- python $R/scripts/png_bench.py > /dev/null 2>&1; cd $R
python profiles/summarize_csv.py stats $o/pngprof 2>/dev/null | grep -i "png\|kernel" | head -8
LILLIPUT_HIP_TRACE=2 timeout 600 python bench.py --workload firehose --steps 2 --warmup 1 --distinct 192 --no-cpu-baseline > $o/firehose.json 2> $o/firehose.err; echo "firehose rc=$?"
grep "item " $o/firehose.err | tail -303 | awk '{k=$4; t[k]+=$10; n[k]++} END {for (k in t) print k, n[k], t[k]/n[k]}'
python - <<PY
import json
d=json.loads(open("$o/firehose.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["verified_outputs_per_format"], d["config"]["verified_identical"])
PY
