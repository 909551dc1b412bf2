#!/bin/bash
# Round-4 measurement set 13: texture-addresser / L1 counters of k_area_420 (is it bound by its 22 dword loads per source row?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/${1:-r04_m13}; mkdir -p $o
export TMPDIR=/tmp LILLIPUT_HIP_STREAMS=1; cd /tmp
B="python $R/bench.py --size 4000 --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 232"
pass() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $o/$n -o $n -- $B > $o/$n.json 2> $o/$n.err || echo "pass $n failed: $(tail -2 $o/$n.err)"; }
pass ta1 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass sq1 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD
cd $R
python - <<PY
import csv, glob, os
from collections import defaultdict
for p in ("ta1","ta2","tcp1","sq1"):
    hits = glob.glob("$o/%s/**/*counter_collection.csv" % p, recursive=True)
    if not hits: print(p, "no csv"); continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(hits[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:32], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for n in sorted({k[0] for k in acc if k[0].startswith("k_area") or k[0].startswith("k_idct") or k[0].startswith("k_huff_write")}):
        print(p, n, {c: round(acc[(n, c)] / max(1, cnt[(n, c)])) for (nn, c) in acc if nn == n})
    for h in hits: os.remove(h)
PY
