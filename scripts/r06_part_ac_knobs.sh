R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_inline; mkdir -p $O
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f)" % (k, x["images_per_s"], x["latency_ms_p50"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for rep in 1 2; do for m in 1 8; do
  LILLIPUT_HIP_DEFER_INLINE_MAX=$m timeout 1000 python bench.py --workload abi --part A --threads 1,2,4,8,16,32,64,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/ka_${m}_$rep.json 2> $O/ka_${m}_$rep.err; show $O/ka_${m}_$rep.json "Part A inline max $m rep $rep"
done; done
