#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2suite; mkdir -p $OUT; cd $REPO
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_full.log
