#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v16; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/scripts/bench_mlp_quad.py > $OUT/q.json 2> $OUT/prof.err
cat $OUT/q.json | tail -1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-200
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
