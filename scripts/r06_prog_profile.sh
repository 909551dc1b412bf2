#!/bin/bash
# usage (GPU box): scripts/r06_prog_profile.sh -> gpurun_out/r06_prog/: kernel trace + SQ counters of the wave-per-scan progressive decoder
# (64 files of 1024 x 1024, one engine), level-by-level and pipelined, and the lanes reference
R=${GRAFT_REPO_ROOT:-$(pwd)}; o=$R/gpurun_out/r06_prog; mkdir -p $o; export TMPDIR=/tmp LILLIPUT_HIP_STREAMS=1; cd /tmp
P="python $R/scripts/r06_prog_one.py 1024 64"
for v in pipelined levels; do
  e=""; [ $v = levels ] && e="LILLIPUT_HIP_PROG_PIPELINE=0"
  env $e timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$v -o t -- $P 1 2 > $o/trace_$v.log 2>&1
  env $e timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $o/sq1_$v -o p -- $P 1 1 > $o/sq1_$v.log 2>&1
  env $e timeout 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $o/sq2_$v -o p -- $P 1 1 > $o/sq2_$v.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_lanes -o t -- python $R/scripts/r06_prog_one.py 1024 64 2 1 > $o/trace_lanes.log 2>&1
cd $R
for v in pipelined levels lanes; do echo "== $v"; grep "mode" $o/trace_$v.log | tail -2; python scripts/r06_trace_list.py $o/trace_$v k_prog 4; done
for v in pipelined levels; do echo "== sq $v"; python scripts/r06_pmc_list.py $o/sq1_$v k_prog_wave 3; python scripts/r06_pmc_list.py $o/sq2_$v k_prog_wave 3; done
