R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_value; mkdir -p $O
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?"; tail -3 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
c=d["config"]
print("value", d["value"], "ms/step", d["ms_per_step"], "| e2e", c.get("end_to_end",{}).get("images_per_s"), "| roofline", d["roofline"]["frac"], d["roofline"].get("achieved"), "| cpu", d.get("cpu_baseline",{}).get("value"), "| ok", c["verified_identical"], c["engines_per_gpu"])
print(c["workload"]); print(c["timed_region"])
PY
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
