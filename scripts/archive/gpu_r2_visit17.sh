#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v17; mkdir -p $OUT; cd $REPO
timeout 1500 python -m pytest tests/test_gpu_events.py -q -x -p no:cacheprovider 2>&1 | tail -30 | tee $OUT/pytest_events.log
