#!/bin/bash
# the headline kernel at saturating ensemble sizes (report the BASELINE size and a saturating size)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
for n in 10000 30000 100000 300000 1000000; do
  st=$(( n >= 300000 ? 6 : 30 ))
  timeout 600 python bench.py --no-cpu-baseline --ntraj $n --steps $st --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(ntraj=d['config']['ntraj_total'], time_segments=d['config']['time_segments'], ms_per_step=round(d['ms_per_step'],4), traj_per_s=round(d['value']), kernel_ms=round(r['kernel_ms'],4), achieved_GBps=round(r['achieved'],1), frac=round(r['frac'],3), forward_ms=d['forward_solve_ms'])))"
done | tee gpurun_out/nsweep.jsonl
