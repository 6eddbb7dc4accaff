#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v4; mkdir -p $OUT; cd $REPO
{
for r in 1 2 3; do
echo "-- plain";  timeout 60 scripts/kbench_ops 10000 13 100 bench 2.6 | tail -1
echo "-- fused final"; KB_FUSED_FINAL=1 timeout 60 scripts/kbench_ops 10000 13 100 bench 2.6 | tail -1
echo "-- graph"; timeout 60 scripts/kbench_ops 10000 13 100 graph 2.6 | tail -1
echo "-- graph + fused final"; KB_FUSED_FINAL=1 timeout 60 scripts/kbench_ops 10000 13 100 graph 2.6 | tail -1
done
for n in 1250 2500; do echo "-- shard $n"; timeout 60 scripts/kbench_ops $n 0 100 bench | tail -1; timeout 60 scripts/kbench_ops $n 0 100 graph | tail -1; KB_FUSED_FINAL=1 timeout 60 scripts/kbench_ops $n 0 100 graph | tail -1; done
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o kb -- $REPO/scripts/kbench_ops 10000 13 50 bench 2.6 > /dev/null 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat $f | cut -c1-200
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
