#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v13; mkdir -p $OUT; cd $REPO
{
for r in 1 2; do for b in g g_kt2; do for a in 2 0; do echo "-- $b alg $a"; timeout 120 scripts/kbench_mlp_$b 4096 150 2 $a | tail -1; done; done; done
} 2>&1 | tee $OUT/mlpbench_kt.log
