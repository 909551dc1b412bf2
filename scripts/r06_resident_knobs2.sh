R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_resident; mkdir -p $O
for rep in 1 2; do for st in 4 6 8; do
    LILLIPUT_HIP_STREAMS=$st timeout 300 python bench.py --distinct 128 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/e_${st}_$rep.json 2> $O/e_${st}_$rep.err
    LILLIPUT_HIP_STREAMS=$st timeout 300 python bench.py --resident --distinct 128 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/r_${st}_$rep.json 2> $O/r_${st}_$rep.err
    LILLIPUT_HIP_STREAMS=$st timeout 300 python bench.py --resident --distinct 128 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --chunk 170 > $O/r170_${st}_$rep.json 2> $O/r170_${st}_$rep.err
    python - <<PY
import json
def v(p):
    try: return json.loads(open(p).read().strip().splitlines()[-1])["value"]
    except Exception as e: return -1
print("engines $st rep $rep: e2e %.1f resident %.1f resident chunk170 %.1f" % (v("$O/e_${st}_$rep.json"), v("$O/r_${st}_$rep.json"), v("$O/r170_${st}_$rep.json")))
PY
done; done
