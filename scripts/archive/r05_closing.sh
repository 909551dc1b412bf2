#!/bin/bash
# Round-5 closing set on one MI355X (multi-symbol counting passes): the GPU suite, smoke, the rocprofv3 kernel statistics + PMC passes
# (scripts/r05_profile.sh), the literal drop-in under concurrent callers (Part A), the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; tag=${1:-r05b}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
if [ -z "$NOPROF" ]; then bash scripts/r05_profile.sh ${tag}_sq; fi
cd $R
if [ -z "$NOABI" ]; then
timeout 600 python bench.py --workload abi --part A --threads 1,8,64,256 --no-cpu-baseline > $O/bench_abi_A.json 2> $O/bench_abi_A.err; echo "abi rc=$?"; tail -c 600 $O/bench_abi_A.json
fi
if [ -z "$NOBENCH" ]; then
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 300 $O/bench_default.err; cat $O/bench_default.json | head -c 1500
fi
