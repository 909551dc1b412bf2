#!/bin/bash
# usage (on the GPU box): scripts/sweep.sh <out-file> "<bench args common>" "<variant args 1>" "<variant args 2>" ...
# Runs bench.py once per variant and appends "variant :: value :: per-kernel ms" lines to gpurun_out/<out-file>.
out=gpurun_out/$1; shift
common=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  python bench.py --no-cpu-baseline $common $v > /tmp/sw.json 2> /tmp/sw.err || { echo "$v :: FAILED $(tail -3 /tmp/sw.err)" >> $out; continue; }
  python - "$v" >> $out <<'PY'
import json,sys
j=json.load(open('/tmp/sw.json'))
pk=j['roofline']['per_kernel']
print(sys.argv[1], '::', j['value'], 'img/s ::', ' | '.join('%s=%.4f'%(k.split(' ')[0],v['ms_per_image']) for k,v in pk.items()), ':: rounds', j['config']['verify_rounds'], 'ok', j['config']['ok_images'], j['config']['first_output_sha256_16'])
PY
done
cat $out
