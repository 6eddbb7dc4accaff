#!/bin/bash
# Round 3, GPU visit 6: wide models with the parameters in LDS; Tsit5 back on the padded stage sum; smoke(); rocprofv3 kernel stats of the bench command.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v6; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_fused.py -q -p no:cacheprovider -x 2>&1 | tail -6 ) | tee $OUT/tests.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee $OUT/smoke.log
timeout 600 python scripts/bench_tsit5.py 10000 2>/dev/null > $OUT/tsit5.jsonl; python -c "
import json
for ln in open('$OUT/tsit5.jsonl'):
    r = json.loads(ln); print('   %-8s %-14s tol %.0e  fwd %.3f (first %.3f)  reverse kernel %.3f ms' % (r['model'], r['alg'], r['abstol'], r['forward_ms'], r['forward_first_call_ms'], r['adjoint_kernel_ms']))
" | tee $OUT/tsit5.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3 | tee $OUT/bench_time.log
python - <<'PY'
import json
r = json.load(open("gpurun_out/r3v6/bench.json"))
print("ms_per_step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "region", r["roofline"]["region_event_ms_per_step"], "fwd", r["forward_solve_ms"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
for s in r.get("shard_sizes", []): print(" shard", s["ntraj"], s["ms_per_step"], s["kernel_ms"], s["implied_speedup_if_allreduce_hidden"])
for c in r.get("other_configs", []): print(" ", c.get("config", "")[:120], "| fwd", c.get("forward_ms"), "| rev", c.get("reverse_ms"), "| kms", c.get("sweep_kernel_ms", c.get("main_kernel_ms")), "| frac", (c.get("roofline") or {}).get("frac"), c.get("error", ""))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/rocprof -o r3 -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
find $OUT/rocprof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}' | tee $OUT/rocprof_kernel_stats_head.txt
