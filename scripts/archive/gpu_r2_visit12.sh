#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v12; mkdir -p $OUT; cd $REPO
{
for r in 1 2 3; do
echo "-- pf4 13"; timeout 60 scripts/kbench_ops_pf4 10000 13 100 bench | tail -1 | cut -c1-260
for s in 13 19 20 26; do echo "-- pf1 $s"; timeout 60 scripts/kbench_ops_pf1 10000 $s 100 bench | tail -1 | cut -c1-260; done
for s in 13 19; do echo "-- pf2 $s"; timeout 60 scripts/kbench_ops_pf2 10000 $s 100 bench | tail -1 | cut -c1-260; done
done
} 2>&1 | tee $OUT/kbench_pf1.log
