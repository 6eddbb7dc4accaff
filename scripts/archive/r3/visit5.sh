#!/bin/bash
# Round 3, GPU visit 5: the whole GPU suite on the current code; wide rows after the shuffle-reduction emitter; Tsit5 stage sums padded vs trimmed (A/B builds).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v5; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -30 ) | tee $OUT/gpu_suite.log
for lib in "" scripts/libhipadj_Tpadded.so ""; do
  echo "== tsit5 library: ${lib:-default (trimmed stage sums)}" | tee -a $OUT/tsit5_ab.log
  HIPADJ_LIBRARY=${lib:+$REPO/$lib} timeout 600 python scripts/bench_tsit5.py 10000 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln)
    if 'error' in r: print('  ', r); continue
    print('   %-8s %-14s tol %.0e  fwd %.3f (first %.3f)  reverse kernel %.3f ms' % (r['model'], r['alg'], r['abstol'], r['forward_ms'], r['forward_first_call_ms'], r['adjoint_kernel_ms']))
" | tee -a $OUT/tsit5_ab.log
done
( time timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3 | tee $OUT/bench_time.log
python - <<'PY'
import json
r = json.load(open("gpurun_out/r3v5/bench.json"))
print("ms_per_step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "region", r["roofline"]["region_event_ms_per_step"], "fwd", r["forward_solve_ms"])
for c in r.get("other_configs", []): print(" ", c.get("config", "")[:120], "| fwd", c.get("forward_ms"), "| rev", c.get("reverse_ms"), "| kms", c.get("sweep_kernel_ms", c.get("main_kernel_ms")), "| frac", (c.get("roofline") or {}).get("frac"), c.get("error", ""))
PY
