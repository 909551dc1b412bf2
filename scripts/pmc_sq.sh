#!/bin/bash
# usage (GPU box): scripts/pmc_sq.sh <tag> [bench args]   -> gpurun_out/<tag>/{trace,sq1,sq2,sq3,fetch,write}/...
# Kernel durations (kernel-trace + stats, no counters) and the SQ / TCC counter passes, each in its own run (gpurun refuses --pmc
# together with the API trace domains; counters serialise the dispatches, so these are EXCLUSIVE per-kernel figures: one stream).
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
export TMPDIR=/tmp LILLIPUT_HIP_STREAMS=1; cd /tmp
B="python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 256 $*"
o=$R/gpurun_out/$tag; mkdir -p $o
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o trace -- $B > $o/trace.json 2> $o/trace.err
pass() { n=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $o/$n -o $n -- $B > $o/$n.json 2> $o/$n.err || echo "pass $n failed: $(tail -2 $o/$n.err)"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA
pass sq3 GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
pass fetch FETCH_SIZE
pass write WRITE_SIZE
ls $o
