R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_lone; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python scripts/r06_one_image.py 512 4096 2>&1 | grep "Part \|Batch"
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f)" % (k, x["images_per_s"], x["latency_ms_p50"]) for k, x in bt.items()), d["config"]["verified_identical"], d["config"]["engine_pool_after"])
PY
}
for part in C A; do
  timeout 1000 python bench.py --workload abi --part $part --threads 1,2,4,8,16,64,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/p_$part.json 2> $O/p_$part.err; show $O/p_$part.json "Part $part"
done
