R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_inline; mkdir -p $O
for rep in 1 2; do for v in 1 0; do
  LILLIPUT_HIP_DEFER_INLINE=$v timeout 1000 python bench.py --workload abi --part A --threads 1,2,4,8,64,256 --batch 6144 --steps 2 --distinct 128 --no-cpu-baseline > $O/a_${v}_$rep.json 2> $O/a_${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/a_${v}_$rep.json").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("inline $v rep $rep:", " | ".join("%s: %.0f img/s p50 %.2f p99 %.1f cpu %.2f" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"], x["host_cpu_ms_per_request"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
done; done
