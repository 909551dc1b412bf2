R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_progb; mkdir -p $O
run() { # size batch streams bound_GiB
  n=$O/q_$1_$2_$3_$4
  LILLIPUT_HIP_PROG_DEVICE_MAX=$(( $4 << 30 )) LILLIPUT_HIP_STREAMS=$3 timeout 900 python bench.py --source-sampling 420p --size $1 --batch $2 --distinct 128 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --end-to-end > $n.json 2> $n.err || tail -3 $n.err
  python - <<PY
import json
try:
    d=json.loads(open("$n.json").read().strip().splitlines()[-1])
    print("size $1 batch $2 engines $3 bound $4 GiB: %.0f img/s, %.1f ms/step, ok %s gate %s" % (d["value"], d["ms_per_step"], d["config"]["ok_images"], d["config"]["verified_identical"]))
except Exception as e: print("size $1 batch $2 engines $3 bound $4: unreadable", e)
PY
}
run 4096 1024 4 4
run 4096 1024 4 8
run 4096 1024 4 16
run 4096 1024 8 8
run 4096 512 4 16
run 1024 2048 4 16
