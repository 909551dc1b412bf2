#!/bin/bash
# PNG path with the library's inflater: parity suite for PNG / firehose, then the firehose bench with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_r}; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests/test_png.py tests/test_png_output.py tests/test_firehose.py tests/test_inflate.py tests/test_webp.py -q -x 2>&1 | tail -6 ) > $o/pytest.log; cat $o/pytest.log
for v in own zlib; do
  e=""; [ $v = zlib ] && e="LILLIPUT_HIP_PNG_ZLIB=1"
  env $e LILLIPUT_HIP_TRACE=2 timeout 600 python bench.py --workload firehose --steps 3 --warmup 1 --distinct 192 --no-cpu-baseline > $o/firehose_$v.json 2> $o/firehose_$v.err; echo "firehose $v rc=$?"
  python - <<PY
import json, re
d=json.loads(open("$o/firehose_$v.json").read().strip().splitlines()[-1]); print("$v", d["value"], d["config"].get("verified_identical"), d["config"].get("per_format"))
ms={}
for line in open("$o/firehose_$v.err"):
    m=re.search(r"item \d+ \((....), \d+ bytes\) on worker \d+: ([0-9.]+) ms", line)
    if m: ms.setdefault(m.group(1),[]).append(float(m.group(2)))
print({k:(len(v), round(sum(v)/len(v),1)) for k,v in ms.items()})
PY
done
python scripts/png_bench.py 2>&1 | tail -8
