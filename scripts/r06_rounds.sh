R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_rounds; mkdir -p $O
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
bt=d["config"]["by_threads"]
print("$2:", " | ".join("%s: %.0f (%.2f / %.1f) redone %s" % (k, x["images_per_s"], x["latency_ms_p50"], x["latency_ms_p99"], x["decode_launches_redone"]) for k, x in bt.items()), d["config"]["verified_identical"])
PY
}
for rep in 1 2; do for r in 4 6; do
  LILLIPUT_HIP_VERIFY_ROUNDS=$r timeout 1000 python bench.py --workload abi --part A --threads 16,64,256 --batch 4096 --steps 2 --distinct 128 --no-cpu-baseline > $O/a_${r}_$rep.json 2> $O/a_${r}_$rep.err; show $O/a_${r}_$rep.json "Part A big-launch rounds $r rep $rep"
done; done
for r in 4 6; do
  LILLIPUT_HIP_VERIFY_ROUNDS=$r timeout 1000 python bench.py --distinct 256 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/h_$r.json 2> $O/h_$r.err
  python - <<PY
import json
d=json.loads(open("$O/h_$r.json").read().strip().splitlines()[-1]); c=d["config"]
print("headline rounds $r: value %.0f e2e %.0f verify_rounds %.2f redone %s gate %s" % (d["value"], c["end_to_end"]["images_per_s"], c["verify_rounds"], c["decode_launches_redone"], c["verified_identical"]))
PY
done
