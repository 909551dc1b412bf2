R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_lone; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python scripts/r06_one_image.py 256 512 1024 2048 4096 2>&1 | grep "Part "
for s in 4096 512; do python scripts/r06_cold_one.py $s 128 A | head -1; python scripts/r06_cold_one.py $s 128 C | head -1; done
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f)" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"], d["config"]["engine_pool_after"])
PY
}
for part in A C; do
  timeout 1000 python bench.py --workload abi --part $part --threads 1,2,4,8,16,64,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/r_$part.json 2> $O/r_$part.err; show $O/r_$part.json "Part $part"
done
