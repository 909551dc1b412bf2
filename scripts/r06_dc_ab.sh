R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_dc; mkdir -p $O; export TMPDIR=/tmp LILLIPUT_HIP_STREAMS=1
for rep in 1 2; do for v in base default; do
  if [ $v = default ]; then unset LILLIPUT_HIP_LIB; else export LILLIPUT_HIP_LIB=$R/lilliput_amd/liblilliput_hip_$v.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v$rep -o t -- python $R/bench.py --resident --no-extra-legs --no-cpu-baseline --distinct 64 --steps 1 --warmup 1 --batch 226 > $O/$v$rep.json 2> $O/$v$rep.err)
  python - <<PY
import csv,glob
f=glob.glob("$O/$v$rep/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if r["Name"].startswith(("k_dc","k_unstuff")): print("$v rep$rep", r["Name"][:24], r["AverageNs"])
PY
done; done
