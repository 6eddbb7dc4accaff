#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v9; mkdir -p $OUT; cd $REPO
{
for r in 1 2 3 4; do
for b in ops ops_pf6 ops_pf4 ops_pf3; do echo "-- $b"; timeout 60 scripts/kbench_$b 10000 13 100 bench | tail -1 | cut -c1-260; done
done
} 2>&1 | tee $OUT/kbench_pf.log
