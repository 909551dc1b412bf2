R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_resident; mkdir -p $O
for rep in 1 2; do for cfg in "4 128" "4 86" "4 256" "6 171" "6 86" "8 128" "8 64" "5 205" "5 103" "7 147"; do set -- $cfg; st=$1; ch=$2
    LILLIPUT_HIP_STREAMS=$st timeout 300 python bench.py --resident --distinct 128 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --chunk $ch > $O/k3_${st}_${ch}_$rep.json 2> $O/k3_${st}_${ch}_$rep.err
    python - <<PY
import json
def v(p):
    try: return json.loads(open(p).read().strip().splitlines()[-1])["value"]
    except Exception as e: return -1
print("engines $st chunk $ch rep $rep: resident %.1f" % v("$O/k3_${st}_${ch}_$rep.json"))
PY
done; done
