#!/bin/bash
# r05: SQ counters of k_png_unfilter on configs[2] (one caller, a few requests): where a step's cycles go
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/png_pmc; rm -rf $O; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --workload png2webp --threads 1 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline"
cd /tmp
pass() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -o $n -- $B > $O/$n.json 2> $O/$n.err || echo "pass $n failed: $(tail -2 $O/$n.err)"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
pass sq3 GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU
cd $O
python3 - <<'PY'
import csv, glob, collections
for n in ("sq1","sq2","sq3"):
    fs = glob.glob(f"{n}/**/*counter_collection.csv", recursive=True)
    if not fs: print(n, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k,row["Counter_Name"])] += 1
    for k in acc:
        if "png_unfilter" in k:
            print(n, k, {c: round(v / max(1,cnt[(k,c)]),1) for c, v in acc[k].items()}, "dispatches", max(cnt[(k,c)] for c in acc[k]))
PY
find $O -name '*.csv' -size +2M -delete
