#!/bin/bash
TAG=${1:-ts5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
echo "== pytest tsit5" ; timeout 900 python -m pytest tests -m gpu -x -q -k "tsit5 or native" 2>&1 | tail -15
echo "== bench tsit5" ; timeout 600 python scripts/bench_tsit5.py 10000 2>&1 | tee $OUT/tsit5.jsonl | cut -c1-330
echo "== full gpu suite" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
