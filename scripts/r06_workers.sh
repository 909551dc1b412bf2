R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_workers; mkdir -p $O
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f)" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for rep in 1 2; do for w in 0 4 8 16; do
  LILLIPUT_HIP_COALESCE_RESIDENT_MAX=$w timeout 1000 python bench.py --workload abi --part A --threads 12,16,24,32,64,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/r_${w}_$rep.json 2> $O/r_${w}_$rep.err; show $O/r_${w}_$rep.json "resident dispatch up to $w rep $rep"
done; done
