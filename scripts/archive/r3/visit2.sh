#!/bin/bash
# Round 3, GPU visit 2: fused tail with 16-byte sc1 buffer ops; changing-data test of the composition tree; cost of the per-kernel events; forward solve unrolled by two.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v2; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 900 python -m pytest tests/test_gpu_fused.py -q -p no:cacheprovider -x 2>&1 | tail -15 ) | tee $OUT/fused_tests.log
run() { # label, env..., args
  echo "== $1" | tee -a $OUT/ab.log; shift
  env "$@" 2>>$OUT/ab.err | tee -a $OUT/ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('   ms_per_step %.5f  kernel_ms %.5f  fwd_ms %.5f  segs %d  parity du0 %.1e dp %.1e' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['forward_solve_ms'], r['config']['time_segments'], r['parity_max_rel_du0_vs_oracle'], r['parity_max_rel_dp_vs_oracle']))
"
}
B="timeout 300 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline"
for nt in 10000 5000 2500 1250; do
  run "ntraj=$nt fused=1 timing=1" HIPADJ_FUSED=1 $B --ntraj $nt --timing 1
  run "ntraj=$nt fused=1 timing=0" HIPADJ_FUSED=1 $B --ntraj $nt --timing 0
  run "ntraj=$nt fused=0 timing=0" HIPADJ_FUSED=0 $B --ntraj $nt --timing 0
done
run "ntraj=10000 fused=1 timing=1 again" HIPADJ_FUSED=1 $B --ntraj 10000 --timing 1
