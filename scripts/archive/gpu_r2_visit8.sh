#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v8; mkdir -p $OUT; cd $REPO
export HIPADJ_NO_TORCH=1
timeout 120 scripts/kbench_fp64 2>&1 | tee $OUT/fp64_issue.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "tsit5" 2>&1 | tail -5 | tee $OUT/pytest_tsit5.log
timeout 600 python scripts/bench_tsit5.py 2>&1 | cut -c1-400 | tee $OUT/tsit5.jsonl
