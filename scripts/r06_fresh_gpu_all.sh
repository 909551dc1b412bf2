#!/bin/bash
# The WHOLE GPU suite on fresh seeds (LILLIPUT_FUZZ_RNG_OFFSET + LILLIPUT_FUZZ_SEED_OFFSET): tests against recorded answers are expected to
# fail where their inputs are seeded -- this run is for triage, not a gate.   scripts/r06_fresh_gpu_all.sh OFFSET...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $R; O=$R/gpurun_out/r06_fresh_all; mkdir -p $O
for k in "$@"; do
  LILLIPUT_FUZZ_RNG_OFFSET=$k LILLIPUT_FUZZ_SEED_OFFSET=$k timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/offset_$k.log 2>&1
  echo "offset $k: $(tail -1 $O/offset_$k.log)"; grep "^FAILED" $O/offset_$k.log | cut -c1-220
done
