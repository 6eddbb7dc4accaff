#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v6; mkdir -p $OUT; cd $REPO
HIPADJ_NO_TORCH=1 timeout 900 python -m pytest tests/test_gpu_at_size.py::test_config5_documented_horizon -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_c5.log; HIPADJ_NO_TORCH=1 timeout 900 python -m pytest tests/test_gpu_at_size.py::test_config5_documented_horizon -m gpu -q -x -p no:cacheprovider --durations=1 2>&1 | tail -4
