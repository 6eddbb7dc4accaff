#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v10; mkdir -p $OUT; cd $REPO
export HIPADJ_NO_TORCH=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "dense_adjoint_record" 2>&1 | tail -6 | tee $OUT/pytest.log
