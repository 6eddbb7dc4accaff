#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v11; mkdir -p $OUT; cd $REPO
timeout 900 python -m pytest tests/test_gpu_mass_matrix.py -q -x -p no:cacheprovider 2>&1 | tail -30 | tee $OUT/pytest_mm.log
