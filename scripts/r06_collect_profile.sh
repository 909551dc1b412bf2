#!/bin/bash
# usage (here, after scripts/r06_profile.sh <tag> ran on the GPU box): scripts/r06_collect_profile.sh <tag> -> profiles/r06_kernel_stats.md, r06_pmc_traffic.json, r06_sq_counters.json
t=${1:-r06_sq}; cd $(dirname $0)/..
{ echo "# Round 6: rocprofv3 --kernel-trace --stats of the headline workload (scripts/r06_profile.sh: bench.py --resident --batch 226, one stream, launches of 113 images of 4096x4096 = one full round of WRITE workgroups; taken on the closing sources: layout switch of lp_huff_core.h, unstuff kernels that request two chunks ahead -- profiles/r06_write_stream.md)"; echo
  grep "^|" gpurun_out/${t}_kernel_stats.md; echo
  echo "## HBM traffic (separate --pmc passes: FETCH_SIZE x 2 = the gfx950 correction, WRITE_SIZE), per image"; echo; grep "^|" gpurun_out/${t}_traffic.md; echo
  echo "## SQ counters (profiles/summarize_sq.py)"; echo; grep "^|" gpurun_out/${t}_sq.md; echo
  echo "## The same trace with launches of 128 images (what an engine of the resident form launches for 1 024 sources since round 6, alone on the GPU: --batch 256 --chunk 128)"; echo
  echo "\`k_huff_write\` 3 448 us per launch (26.9 us per image against 19.5): 1 152 workgroups are a full round of 1 024 and a nearly empty second one. Alone that launch size loses; with eight engines the other engines' kernels fill the second round and the set runs at 21.1 k images/s against 19.5 k (profiles/r06_resident.md). The exclusive figures above stay on full rounds."
} > profiles/r06_kernel_stats.md
cp gpurun_out/${t}_pmc_traffic.json profiles/r06_pmc_traffic.json; cp gpurun_out/${t}_sq.json profiles/r06_sq_counters.json
python - <<PY
import json,sys
sys.path.insert(0,'.')
import bench
print("stamp", bench.kernel_source_sha16(), json.load(open('profiles/r06_pmc_traffic.json'))['kernel_source_sha16'])
PY
