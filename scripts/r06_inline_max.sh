R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_imax; mkdir -p $O
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f)" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for rep in 1 2 3; do for q in 0 1; do
  LILLIPUT_HIP_LONE_PRIORITY=$q timeout 1000 python bench.py --workload abi --part A --threads 8,6,4,2 --batch 2048 --steps 2 --distinct 128 --no-cpu-baseline > $O/p_${q}_$rep.json 2> $O/p_${q}_$rep.err; show $O/p_${q}_$rep.json "LONE_PRIORITY=$q rep $rep"
done; done
