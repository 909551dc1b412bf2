R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_inline; mkdir -p $O
for rep in 1 2; do for rule in serving count; do
  LILLIPUT_HIP_COALESCE_RULE=$rule timeout 1000 python bench.py --workload abi --part C --threads 1,2,4,8,64,256 --batch 6144 --steps 2 --distinct 128 --no-cpu-baseline > $O/c_${rule}_$rep.json 2> $O/c_${rule}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/c_${rule}_$rep.json").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("Part C rule $rule rep $rep:", " | ".join("%s: %.0f img/s p50 %.2f p99 %.1f" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
done; done
