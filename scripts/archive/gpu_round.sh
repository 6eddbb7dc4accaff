#!/bin/bash
# One gpurun visit: smoke, GPU parity tests, headline bench, rocprofv3 kernel trace (+ optional PMC passes).
# usage: gpurun --timeout 1800 -- 'bash scripts/gpu_round.sh <tag> [pmc]'
TAG=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench" ; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err ; tail -3 $OUT/bench.err ; cat $OUT/bench.json
for seg in 1 6 13; do
  echo "== bench segments=$seg" ; timeout 300 python bench.py --no-cpu-baseline --segments $seg --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'])"
done
cd /tmp ; export TMPDIR=/tmp
echo "== rocprofv3 kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --no-cpu-baseline --steps 20 > $OUT/prof_bench.json 2> $OUT/prof.err
find $OUT/prof -name "*stats*" | head ; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1) ; [ -n "$f" ] && head -12 $f
if [ "$2" == "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== rocprofv3 pmc $c"
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_$c.err
    f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r.get("Kernel_Name", "?")[:60], r.get("Counter_Name"))
    agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:8]:
    print(f"{k:60s} {c:12s} total={v:.4g} launches={n} per_launch={v/n:.4g}")
PY
  done
fi
# keep only the summaries (the raw traces can be large)
find $OUT -name "*.csv" -size +2M -delete
echo "== done"
