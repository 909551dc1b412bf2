#!/bin/bash
# round 3, GPU call 1: the whole GPU test suite, the default bench line (zero-copy ingest), the other ingest routes on the same cached
# sources, the host-side ingest rehearsal for 1/2/4/8 engine sets and a one-rank RCCL smoke of bench.py's multi-GPU path.
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r03_a; mkdir -p $o
export TMPDIR=/tmp; cd $R
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $o/pytest.log; cat $o/pytest.log
timeout 900 python bench.py --steps 10 --warmup 2 > $o/bench_auto.json 2> $o/bench_auto.err; echo "bench auto rc=$?"; tail -c 300 $o/bench_auto.err
timeout 300 python bench.py --steps 10 --warmup 2 --ingest pinned --no-extra-legs --no-cpu-baseline > $o/bench_pinned.json 2> $o/bench_pinned.err; echo "bench pinned rc=$?"
timeout 300 python bench.py --steps 10 --warmup 2 --ingest staged --no-extra-legs --no-cpu-baseline > $o/bench_staged.json 2> $o/bench_staged.err; echo "bench staged rc=$?"
LILLIPUT_HIP_NUMA=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline > $o/bench_auto_nonuma.json 2> $o/bench_auto_nonuma.err; echo "bench auto (no NUMA binding) rc=$?"
LD_LIBRARY_PATH=$R/lilliput_amd timeout 200 $R/scripts/ingest_scale /tmp/lilliput_bench_4096_q90 2.0 > $o/ingest_scale.md 2> $o/ingest_scale.err; echo "ingest_scale rc=$?"; cat $o/ingest_scale.md
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline > $o/bench_nccl1.json 2> $o/bench_nccl1.err; echo "one-rank nccl rc=$?"; tail -c 300 $o/bench_nccl1.err
for f in auto pinned staged auto_nonuma nccl1; do python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["config"].get("ingest"), "verified", d["config"].get("verified_outputs"), d["config"].get("verified_identical"), "h2d", d["config"].get("h2d_GBps_per_rank"))
except Exception as e: print("$f", "no json", e)
PY
done
numactl -H 2>/dev/null | head -20; lscpu | grep -i "numa\|model name\|socket" ; cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c
