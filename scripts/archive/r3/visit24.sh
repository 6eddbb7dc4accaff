#!/bin/bash
# GPU visit 24: tree radix 4 vs 8 on the strong-scaling shards (16-byte sc1 hand-offs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v24; O=gpurun_out/r3v24
for rep in 1 2; do for n in 1250 2500; do for r in 4 8; do
  HIPADJ_TREE_RADIX=$r timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 --ntraj $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ntraj $n radix $r rep $rep ms_per_step %.5f kernel_ms %.5f segs %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['time_segments']))" | tee -a $O/radix.log
done; done; done
