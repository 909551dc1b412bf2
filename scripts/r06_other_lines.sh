R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_other; mkdir -p $O
for w in png2webp animated firehose; do
  timeout 1200 python bench.py --workload $w > $O/$w.json 2> $O/$w.err; echo "$w rc $?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/$w.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$w: value %.1f %s | cpu %s | gate %s | roofline %s" % (d["value"], d["unit"], (d.get("cpu_baseline") or {}).get("value"), c.get("verified_identical", c.get("verified")), (d.get("roofline") or {}).get("bound")))
except Exception as e: print("$w unreadable", e)
PY
done
timeout 900 python bench.py --size 4000 --distinct 256 --steps 5 --no-cpu-baseline > $O/s4000.json 2> $O/s4000.err
timeout 900 python bench.py --source-quality 75 --distinct 256 --steps 5 --no-cpu-baseline > $O/q75.json 2> $O/q75.err
timeout 900 python bench.py --restart-rows 1 --distinct 256 --steps 5 --no-cpu-baseline > $O/dri.json 2> $O/dri.err
for f in s4000 q75 dri; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); c=d["config"]
print("$f: value %.0f | e2e %.0f | frac %.3f | gate %s" % (d["value"], c["end_to_end"]["images_per_s"], d["roofline"]["frac"], c["verified_identical"]))
PY
done
