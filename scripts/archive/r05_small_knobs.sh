#!/bin/bash
# Round 5: small sources end to end against the chunk size and the number of engines:
#   bash scripts/r05_small_knobs.sh <size> <batch> cfg...   cfg = streams:chunk (chunk 0 = automatic)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r05_small_knobs; mkdir -p $O
sz=${1:-512}; batch=${2:-2048}; shift; shift
for cfg in "$@"; do
    st=${cfg%%:*}; ch=${cfg##*:}
    export LILLIPUT_HIP_STREAMS=$st
    if [ $ch = 0 ]; then unset LILLIPUT_HIP_PIPE_CHUNK; else export LILLIPUT_HIP_PIPE_CHUNK=$ch; fi
    timeout 300 python bench.py --distinct 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --size $sz --batch $batch > $O/b_${sz}_${batch}_${st}_$ch.json 2> $O/b_${sz}_${batch}_${st}_$ch.err || tail -3 $O/b_${sz}_${batch}_${st}_$ch.err
    python - <<PY
import json
d=json.loads(open("$O/b_${sz}_${batch}_${st}_$ch.json").read().strip().splitlines()[-1])
print("size $sz batch $batch streams $st chunk $ch", "e2e", d["value"], "ms/step", d["ms_per_step"])
PY
done
