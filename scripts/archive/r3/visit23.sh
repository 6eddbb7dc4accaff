#!/bin/bash
# Round 3, last visit: the adaptive / runtime-model tests after the opt-in revert, then the bench line and its rocprofv3 kernel trace on the round's last code
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v23; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "tsit5 or Tsit5 or adaptive or runtime or fuzz_mm" > $OUT/ts5_runtime_tests.log 2>&1; tail -3 $OUT/ts5_runtime_tests.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err )
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; rm -rf $OUT/trace
cd $REPO; python - <<'PY'
import json, csv
r = json.loads(open("gpurun_out/r3v23/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"], "whole", r["roofline"]["whole_pass_frac"], "fwd", r["forward_solve_ms"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
for s in r.get("shard_sizes", []): print(" shard", s["ntraj"], s["ms_per_step"], s["kernel_ms"], s["implied_speedup_if_allreduce_hidden"])
for o in r.get("other_configs", []): print("  ", o["config"][:120], "| rev", o.get("reverse_ms"), "| frac", (o.get("roofline") or {}).get("frac"))
for row in csv.reader(open("gpurun_out/r3v23/kernel_stats.csv")): print(row[0][:60], row[1:7])
PY
