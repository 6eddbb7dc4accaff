#!/bin/bash
# Round 3, final evidence visit: whole -m gpu suite, smoke, the bench line, rocprofv3 kernel trace of the same command, PMC traffic of the dominant kernel
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v20; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $OUT/gpu_suite.log 2>&1; tail -14 $OUT/gpu_suite.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err )
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -4 $OUT/kernel_stats.csv | cut -c1-60,300-420
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
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cd $REPO; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r3v20/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"], "fwd", r["forward_solve_ms"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
for s in r.get("shard_sizes", []): print(" shard", s["ntraj"], s["ms_per_step"], s["kernel_ms"], s["implied_speedup_if_allreduce_hidden"])
for o in r.get("other_configs", []): print("  ", o["config"][:140], "| rev", o.get("reverse_ms"), "| frac", (o.get("roofline") or {}).get("frac"))
PY
