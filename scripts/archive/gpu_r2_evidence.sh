#!/bin/bash
# round 2 evidence visit: full GPU suite, smoke, bench line, rocprofv3 kernel trace of the same bench command, PMC traffic passes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2ev; mkdir -p $OUT; cd $REPO
if [ -z "$HIPADJ_EVIDENCE_SKIP_PYTEST" ]; then echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_full.log; fi
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; head -c 1500 $OUT/bench.json; echo
cd /tmp; export TMPDIR=/tmp
echo "== rocprofv3 kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 20 > $OUT/prof_bench.json 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-260
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 pmc $c"
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $OUT/pmc_hbm.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r.get("Kernel_Name", "?")[:70], r.get("Counter_Name"))
    agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:6]:
    print(f"{k:70s} {c:12s} total={v:.6g} launches={n} per_launch={v/n:.6g}")
PY
done
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
echo "== done"
