#!/bin/bash
# GPU visit 14: weight of the top segment (HIPADJ_WTOP) for the one-launch kernel at 10^4 and on the shards
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v14; O=gpurun_out/r3v14
for rep in 1 2; do for w in 2.6 2.2 3.0 3.4; do
  HIPADJ_WTOP=$w timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wtop $w rep $rep ntraj 10000 ms_per_step %.5f kernel_ms %.5f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $O/wtop.log
done; done
for w in 2.6 2.2 3.0; do
  HIPADJ_WTOP=$w timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 --ntraj 1250 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wtop $w ntraj 1250 ms_per_step %.5f kernel_ms %.5f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $O/wtop.log
done
