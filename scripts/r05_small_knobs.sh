#!/bin/bash
# Round 5: small sources end to end against the number of engines and the chunk size
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r05_small_knobs; mkdir -p $O
sz=${1:-512}
for cfg in "4:0" "8:0" "4:64" "4:256" "8:256" "6:0"; do
    st=${cfg%%:*}; ch=${cfg##*:}
    export LILLIPUT_HIP_STREAMS=$st
    if [ $ch = 0 ]; then unset LILLIPUT_HIP_PIPE_CHUNK; else export LILLIPUT_HIP_PIPE_CHUNK=$ch; fi
    timeout 300 python bench.py --distinct 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --size $sz --batch 4096 > $O/b_${st}_$ch.json 2> $O/b_${st}_$ch.err || tail -3 $O/b_${st}_$ch.err
    python - <<PY
import json
d=json.loads(open("$O/b_${st}_$ch.json").read().strip().splitlines()[-1])
print("size $sz streams $st chunk $ch", "e2e", d["value"], "ms/step", d["ms_per_step"])
PY
done
