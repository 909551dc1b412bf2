R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_workers; mkdir -p $O
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f)" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for rep in 1 2 3; do for w in 0 4; do
  LILLIPUT_HIP_COALESCE_EXTRA=$w timeout 1000 python bench.py --workload abi --part A --threads 32,64,96,128,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/x_${w}_$rep.json 2> $O/x_${w}_$rep.err; show $O/x_${w}_$rep.json "extra dispatchers $w rep $rep"
done; done
