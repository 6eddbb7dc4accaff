#!/bin/bash
# evidence refresh without the test suite: bench line (with cpu_baseline) + rocprofv3 kernel stats of the same command
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/final; mkdir -p $OUT; cd $REPO
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-400
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --no-cpu-baseline --steps 20 > $OUT/prof_bench.json 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-70,180-300
find $OUT -name "*.csv" -size +2M -delete
