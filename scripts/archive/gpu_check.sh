#!/bin/bash
# quick visit: tests + headline bench + per-algorithm timings. usage: gpurun -- 'bash scripts/gpu_check.sh <tag>'
TAG=${1:-chk}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== bench" ; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err ; tail -2 $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print('traj/s %.4e ms/step %.4f kernel_ms %.4f frac %.3f  parity %.2e cpu %.3e'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'],d['parity_max_rel_du0_vs_oracle_sample'],d['cpu_baseline']['value']))"
echo "== per-algorithm" ; timeout 600 python scripts/bench_algs.py 10000 2>/dev/null | tee $OUT/algs.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['alg'], d['kw'], 'segs',d['time_segments'],'wall_ms %.4f kernel_ms %.4f traj/s %.3e'%(d['wall_ms'],d['main_kernel_ms'],d['traj_per_s']))"
echo "== done"
