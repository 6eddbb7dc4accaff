#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v13; mkdir -p $OUT; cd $REPO
{
for r in 1 2; do
for b in s0e0 k4s0 k4s1 k8s0 k8s1 k8s1e1; do echo "-- $b gauss"; timeout 120 scripts/kbench_mlp_$b 4096 150 3 2 | tail -1; done
done
for b in k8s1; do echo "-- $b interp"; timeout 120 scripts/kbench_mlp_$b 4096 150 3 0 | tail -1; done
} 2>&1 | tee $OUT/mlpbench.log
