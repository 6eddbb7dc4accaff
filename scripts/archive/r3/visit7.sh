#!/bin/bash
# Round 3, GPU visit 7: whole GPU suite after the runtime lane models moved to the one-launch pass; PMC traffic of the fused kernel; bench line.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v7; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=8 2>&1 | tail -22 ) | tee $OUT/gpu_suite.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == c: acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k_interp" in k or "k_forward" in k: print(c, k, "launches", len(v), "mean per launch", sum(v) / len(v))
PY
done
cd $REPO
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -6 $OUT/kernel_stats.csv
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ); python - <<'PY'
import json
r = json.load(open("gpurun_out/r3v7/bench.json"))
print("ms_per_step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "fwd", r["forward_solve_ms"])
for s in r.get("shard_sizes", []): print(" shard", s["ntraj"], s["ms_per_step"], s["kernel_ms"], s["implied_speedup_if_allreduce_hidden"])
PY
