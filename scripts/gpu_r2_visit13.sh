#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v13; mkdir -p $OUT; cd $REPO
{
scripts/kbench_tanh
for r in 1 2; do for b in g g_ft; do echo "-- $b gauss"; timeout 120 scripts/kbench_mlp_$b 4096 150 2 2 | tail -1; done; done
} 2>&1 | tee $OUT/mlpbench_tanh.log
