#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v14; mkdir -p $OUT; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -k "mlp or config4 or C4" -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_mlp.log
