# what the driver runs at round end, on the committed tree: the GPU tests, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_rehearsal; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print("value %.1f %s | ms_per_step %.2f | e2e %.1f | roofline frac %.3f achieved %.0f traffic %s (x%s) | cpu_baseline %s on %s cores | gate %s (%d outputs) | redone %s" % (
    d["value"], d["unit"], d["ms_per_step"], c["end_to_end"]["images_per_s"], r["frac"], r["achieved"], r["traffic"], r["traffic_over_algorithmic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"],
    c["verified_identical"], c["verified_outputs"], c["decode_launches_redone"]))
PY
