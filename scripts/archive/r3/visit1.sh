#!/bin/bash
# Round 3, GPU visit 1: the one-launch reverse pass (HIPADJ_FUSED) and the event-list forward solve (HIPADJ_FWD_EV) against the round-2 forms:
# parity (at-size tests + the Interpolating part of the parity suite), A/B timings through bench.py at the ensemble and shard sizes, CPU scaling study.
#   gpurun --timeout 1500 -- 'bash scripts/r3/visit1.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v1; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 600 python -m pytest tests/test_gpu_at_size.py -q -p no:cacheprovider -x -k "config2 or config3" 2>&1 | tail -5 ) | tee $OUT/at_size.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "interp or segment or Interpolating or full_size or torch or save_idxs" 2>&1 | tail -8 ) | tee $OUT/parity_subset.log
for nt in 10000 5000 2500 1250; do
  for v in "0 4" "1 4" "1 8"; do
    set -- $v
    echo "== ntraj=$nt fused=$1 radix=$2" | tee -a $OUT/ab.log
    HIPADJ_FUSED=$1 HIPADJ_TREE_RADIX=$2 timeout 300 python bench.py --ntraj $nt --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>>$OUT/ab.err | tee -a $OUT/ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('   ms_per_step %.5f  kernel_ms %.5f  fwd_ms %.5f  segs %d  parity du0 %.1e dp %.1e' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['forward_solve_ms'], r['config']['time_segments'], r['parity_max_rel_du0_vs_oracle'], r['parity_max_rel_dp_vs_oracle']))
"
  done
done
echo "== forward per-knot form (HIPADJ_FWD_EV=0)" | tee -a $OUT/ab.log
HIPADJ_FWD_EV=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>>$OUT/ab.err | tee -a $OUT/ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('   fwd_ms %.5f' % r['forward_solve_ms'])
"
# CPU baseline scaling study on this host (oracle, OpenMP): thread counts x binding policies
python - <<'PY' 2>&1 | tee $OUT/cpu_scaling.log
import os, sys, time, subprocess, json
code = r'''
import os, sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, bench
u0, p = bench.inputs(10000)
pr = bench.oracle_problem()
nt = int(sys.argv[1])
n = (len(u0) // nt) * nt
pr.adjoint_ensemble(u0[:n], p, nthreads=nt, want_out=False)
r = []
for _ in range(5):
    t0 = time.perf_counter(); _, _, _, tm = pr.adjoint_ensemble(u0[:n], p, nthreads=nt, want_out=False); w = time.perf_counter() - t0
    r.append((n / tm["reverse_s"], n / w))
r.sort()
print(json.dumps(dict(threads=nt, bind=os.environ.get("OMP_PROC_BIND"), places=os.environ.get("OMP_PLACES"), rev_rate_med=r[2][0], rev_rate_min=r[0][0], rev_rate_max=r[4][0], wall_rate_med=sorted(x[1] for x in r)[2])))
'''
print("host threads", os.cpu_count())
for bind, places in (("close", "cores"), ("spread", "threads"), ("false", "")):
    for nt in (1, 8, 16, 32, 64, 128, 256):
        if nt > (os.cpu_count() or 1): continue
        env = dict(os.environ, OMP_PROC_BIND=bind)
        if places: env["OMP_PLACES"] = places
        else: env.pop("OMP_PLACES", None)
        try:
            print(subprocess.run([sys.executable, "-c", code, str(nt)], env=env, capture_output=True, text=True, timeout=120).stdout.strip(), flush=True)
        except Exception as e:
            print("fail", nt, bind, e)
PY
lscpu | head -25 > $OUT/lscpu.txt
