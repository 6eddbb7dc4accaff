#!/bin/bash
# headline bench + rocprofv3 kernel stats of the same command: roofline.kernel_ms (dispatch events) must agree with the trace
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/evt; mkdir -p $OUT; cd $REPO
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --no-cpu-baseline --steps 50 > $OUT/prof_bench.json 2> $OUT/prof.err
python -c "import json; d=json.load(open('$OUT/prof_bench.json')); print('under rocprof: bench kernel_ms', d['roofline']['kernel_ms'])"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-60,200-300
find $OUT -name "*.csv" -size +2M -delete
