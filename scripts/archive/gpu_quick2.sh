#!/bin/bash
# torch-free timing visit: headline + off-grid sweep, plain and under rocprofv3 --kernel-trace --stats
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/quick2; mkdir -p $OUT; cd $REPO
export HIPADJ_NO_TORCH=1
timeout 60 python scripts/bench_notorch.py > $OUT/bench_notorch.jsonl 2> $OUT/bench_notorch.err; cat $OUT/bench_notorch.jsonl | cut -c1-330
cd /tmp; export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/scripts/bench_notorch.py > $OUT/prof_bench.jsonl 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-60,150-260
find $OUT -name "*.csv" -size +2M -delete
