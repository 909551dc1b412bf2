R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_lone; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f)" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for part in A C; do
  timeout 1000 python bench.py --workload abi --part $part --threads 1,2,4,8,16,64,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/s_$part.json 2> $O/s_$part.err; show $O/s_$part.json "Part $part"
done
