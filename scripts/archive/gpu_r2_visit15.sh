#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v15; mkdir -p $OUT; cd $REPO
timeout 1200 python -m pytest tests -m gpu -q -k "bruss or config5 or C5 or horizon or field" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_bruss.log
timeout 300 python scripts/bench_bruss.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bruss_T1024.jsonl
