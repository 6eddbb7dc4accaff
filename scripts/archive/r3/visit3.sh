#!/bin/bash
# Round 3, GPU visit 3: wide runtime models (first run), segment counts of the one-launch pass at the shard sizes, the whole bench line.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v3; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 1200 python -m pytest tests/test_gpu_wide.py -q -p no:cacheprovider -x 2>&1 | tail -25 ) | tee $OUT/wide_tests.log
run() { echo "== $1" | tee -a $OUT/seg.log; shift
  "$@" 2>>$OUT/seg.err | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('   ms_per_step %.5f  kernel_ms %.5f  segs %d  launches %s  parity dp %.1e' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['time_segments'], r['roofline']['launches_per_pass'], r['parity_max_rel_dp_vs_oracle']))
" | tee -a $OUT/seg.log; }
B="timeout 300 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline"
for s in 0 40 64 80 102; do run "ntraj=1250 segments=$s" $B --ntraj 1250 --segments $s; done
for s in 0 44 60; do run "ntraj=2500 segments=$s" $B --ntraj 2500 --segments $s; done
for s in 0 20 23 24 26; do run "ntraj=5000 segments=$s" $B --ntraj 5000 --segments $s; done
for s in 0 12; do run "ntraj=10000 segments=$s" $B --ntraj 10000 --segments $s; done
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3 | tee $OUT/bench_time.log
head -c 1500 $OUT/bench.json; echo
