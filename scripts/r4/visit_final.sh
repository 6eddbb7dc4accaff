#!/bin/bash
# Round 4, final evidence visit: whole -m gpu suite, smoke, the bench line, rocprofv3 kernel stats of the same command, the micro-benchmarks the docs cite
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r4final; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $OUT/gpu_suite.log 2>&1; tail -14 $OUT/gpu_suite.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ); tail -c 300 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv; rm -rf $OUT/trace
python $REPO/scripts/r4/phases.py $OUT/kernel_trace.csv 20 5 | tee $OUT/kernel_phases.txt
cd $REPO
scripts/r4/quad_fwd > $OUT/quad_fwd.log 2>&1
( HIPADJ_QUAD=1 python scripts/r4/ts5_bench.py; HIPADJ_QUAD=0 python scripts/r4/ts5_bench.py ) 2>&1 | grep -v amdgpu.ids > $OUT/ts5_quad_ab.log
python scripts/r4/node_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/node_bench.log
python - <<'PY'
import json, csv
r = json.loads(open("gpurun_out/r4final/bench.json").read().strip().splitlines()[-1])
rf = r["roofline"]
print("ms_per_step", r["ms_per_step"], "value", r["value"], "kernel_ms", rf["kernel_ms"], "frac", rf["frac"], "traffic", rf["traffic"], "cold", r["cold_burst"], "fwd", r["forward_solve_ms"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
for s in r.get("shard_sizes", []): print(" shard", s["ntraj"], s["ms_per_step"], s["implied_speedup_if_allreduce_hidden"])
print(" sat", r.get("saturating_ensemble"))
for row in list(csv.reader(open("gpurun_out/r4final/kernel_stats.csv")))[:5]: print(row[0][:60], row[1:7])
PY
