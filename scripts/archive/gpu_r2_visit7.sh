#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v7; mkdir -p $OUT; cd $REPO
export HIPADJ_NO_TORCH=1
timeout 300 python scripts/bench_notorch.py 2>&1 | cut -c1-330 | tee $OUT/bench_notorch.jsonl
for w in 1.8 2.1 2.7; do echo "wtop $w"; HIPADJ_WTOP=$w timeout 300 python scripts/bench_notorch.py 2>&1 | grep "time-segmented k_offgrid_seg" | cut -c1-330; done
