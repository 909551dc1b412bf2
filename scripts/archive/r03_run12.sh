#!/bin/bash
# tunables of the decode kernels after the step rewrite (flush period, ring size), and the latency of a small call
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r03_o}; mkdir -p $o
export TMPDIR=/tmp; cd $R
for v in base f3 f6 r16; do
  lib=$R/lilliput_amd/liblilliput_hip_$v.so; [ $v = base ] && lib=$R/lilliput_amd/liblilliput_hip.so
  LILLIPUT_HIP_LIB=$lib timeout 400 python bench.py --steps 6 --warmup 2 --distinct 256 --no-cpu-baseline > $o/bench_$v.json 2> $o/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", "e2e", d["value"], "resident", d["config"].get("resident_images_per_s"), "verified", d["config"].get("verified_identical"), d["roofline"]["per_kernel_exclusive_us_per_image"])
except Exception as e: print("$v", "no json", e); print(open("$o/bench_$v.err").read()[-500:])
PY
done
# a small call: 8 / 16 images of 4096 x 4096 through lilliput_hip_batch_transform, subsequence floor 1024 (default) against 4096
python - <<'PY'
import os, sys, time, subprocess
code = r'''
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from lilliput_amd import synth, binding as la
n = int(sys.argv[1])
srcs = [synth.synth_jpeg(i, 4096) for i in range(n)]
b = la.Batch(0)
for rep in range(4):
    t = time.time(); res = b.transform(srcs, 256, 256, quality=85); dt = (time.time() - t) * 1e3
    assert all(r.status == 0 for r in res)
print("n=%d MIN_S=%s: %.2f ms per call" % (n, os.environ.get("LILLIPUT_HIP_MIN_S", "default"), dt))
'''
for n in (4, 8, 16):
    for ms in (None, "2048", "4096"):
        env = dict(os.environ)
        if ms: env["LILLIPUT_HIP_MIN_S"] = ms
        r = subprocess.run([sys.executable, "-c", code, str(n)], env=env, capture_output=True, text=True, timeout=300)
        print((r.stdout.strip() or r.stderr[-400:]))
PY
