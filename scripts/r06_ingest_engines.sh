R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_ingest_chunk; mkdir -p $O
for rep in 1 2; do for q in 75; do for cfg in "4 0" "6 0" "8 0" "6 64" "8 64" "5 0"; do set -- $cfg; st=$1; ch=$2
    LILLIPUT_HIP_STREAMS=$st timeout 300 python bench.py --distinct 128 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --source-quality $q --chunk $ch > $O/e_${q}_${st}_${ch}_$rep.json 2> $O/e_${q}_${st}_${ch}_$rep.err
    python - <<PY
import json
def v(p):
    try: return json.loads(open(p).read().strip().splitlines()[-1])["value"]
    except Exception as e: return -1
print("q$q engines $st chunk %3s rep $rep: e2e %.1f" % ("$ch", v("$O/e_${q}_${st}_${ch}_$rep.json")))
PY
done; done; done
