R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_progl; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { # size batch
  n=$O/line_$1_$2
  timeout 1500 python bench.py --source-sampling 420p --size $1 --batch $2 --distinct 128 --steps 3 --warmup 1 > $n.json 2> $n.err || tail -3 $n.err
  python - <<PY
import json
try:
    d=json.loads(open("$n.json").read().strip().splitlines()[-1])
    c=d["config"]
    print("size $1 batch $2: value %.0f img/s (%.1f ms/step) | e2e %s | ok %s gate %s | cpu %s" % (d["value"], d["ms_per_step"], c.get("end_to_end",{}).get("images_per_s"), c["ok_images"], c["verified_identical"], (d.get("cpu_baseline") or {}).get("value")))
except Exception as e: print("size $1 batch $2: unreadable", e)
PY
}
run 1024 2048
run 4096 1024
