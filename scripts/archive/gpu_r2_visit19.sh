#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v19; mkdir -p $OUT; cd $REPO
{
for r in 1 2 3; do
  for v in 1 0; do
    echo "-- compose16=$v"; HIPADJ_COMPOSE16=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['kernel_ms'], d.get('parity_max_rel_dp_vs_oracle'))"
  done
done
for v in 1 0; do echo "-- compose16=$v ntraj 1250"; HIPADJ_COMPOSE16=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 --ntraj 1250 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['kernel_ms'])"; done
} 2>&1 | tee $OUT/compose16.log
