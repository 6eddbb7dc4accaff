#!/bin/bash
# GPU visit 15: fewer, longer segments (less redundant column work, one wave per SIMD) vs the planner's 13 at 10^4 trajectories
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v15; O=gpurun_out/r3v15
for rep in 1 2; do for c in 0 6 5 8 10 4; do
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 --segments $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('segments $c rep $rep ms_per_step %.5f kernel_ms %.5f segs %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['time_segments']))" | tee -a $O/segs.log
done; done
