R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_hwq; mkdir -p $O
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f)" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for rep in 1 2; do for q in 0 16; do
  if [ $q = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 1000 python bench.py --workload abi --part A --threads 2,8,16,64,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/a_${q}_$rep.json 2> $O/a_${q}_$rep.err; show $O/a_${q}_$rep.json "Part A GPU_MAX_HW_QUEUES=$q rep $rep"
  timeout 1000 python bench.py --distinct 256 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/h_${q}_$rep.json 2> $O/h_${q}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/h_${q}_$rep.json").read().strip().splitlines()[-1]); c=d["config"]
print("headline GPU_MAX_HW_QUEUES=$q rep $rep: value %.0f e2e %.0f gate %s" % (d["value"], c["end_to_end"]["images_per_s"], c["verified_identical"]))
PY
done; done
