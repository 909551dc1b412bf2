R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_ismall; mkdir -p $O
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f)" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for size in 512 1024; do for m in 8 16 32; do
  LILLIPUT_HIP_DEFER_INLINE_MAX=$m timeout 1000 python bench.py --workload abi --part A --size $size --threads 8,12,16,24,32,64 --batch 8192 --steps 2 --distinct 256 --no-cpu-baseline > $O/s_${size}_$m.json 2> $O/s_${size}_$m.err; show $O/s_${size}_$m.json "size $size inline max $m"
done; done
