#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v20; mkdir -p $OUT; cd $REPO
timeout 1500 python -m pytest tests/test_gpu_fuzz_mm_events.py -q -p no:cacheprovider -k "28 or 39" 2>&1 | grep -E "AssertionError: \{|where .* = rel|passed|failed" | cut -c1-600 | tee $OUT/pytest_fuzz2.log
