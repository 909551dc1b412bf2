R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_hot; mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
    if [ $v = default ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
    timeout 400 python bench.py --resident --distinct 128 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err || tail -5 $O/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${v}_$rep.json").read().strip().splitlines()[-1])
    r=d["roofline"]; pk=r["per_kernel_exclusive_us_per_image"]
    ent=sum(v for k,v in pk.items() if k.startswith(("k_huff","k_unstuff")))
    print("%-10s rep$rep resident %8.1f ok %s | write %.2f spec %.2f verify %.2f unstuff %.2f idct %.2f | entropy %.2f all %.2f" % ("$v", d["value"], d["config"].get("ok_images"),
          pk.get("k_huff_write",0), pk.get("k_huff_spec",0), pk.get("k_huff_verify",0), pk.get("k_unstuff_*",0), pk.get("k_idct",0), ent, sum(pk.values())))
except Exception as e:
    print("$v rep$rep unreadable", e)
PY
done
done
unset LILLIPUT_HIP_LIB
