R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_lone; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do
  echo "== rep $rep"; timeout 600 python scripts/r06_one_image.py 256 512 1024 2048 4096 2>&1 | grep "Part \|Batch"
done
cd /tmp; export TMPDIR=/tmp
for side in 512; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$side -o t -- python $R/scripts/r06_one_parta.py $side 60 > $O/trace_$side.log 2>&1
  python $R/scripts/r06_trace_list.py $O/trace_$side "" 30 > $O/trace_${side}_list.txt 2>&1
  find $O/trace_$side -name "*.csv" -size +2M -delete
done
cat $O/trace_512_list.txt
