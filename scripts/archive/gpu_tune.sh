#!/bin/bash
# tuning visit: tests, segment/w_top sweep, SQ counters.  usage: gpurun -- 'bash scripts/gpu_tune.sh <tag>'
TAG=${1:-tune}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
one() { python bench.py --no-cpu-baseline --segments $1 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg',d['config']['time_segments'],'wtop','$HIPADJ_WTOP','traj/s %.3e ms/step %.4f kernel_ms %.4f GB/s %.0f'%(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved']))"; }
echo "== sweep"
for w in 2.0 2.8 3.5 4.5; do for seg in 6 12 13; do HIPADJ_WTOP=$w one $seg; done; done
one 1; one 0
cd /tmp ; export TMPDIR=/tmp
echo "== kernel trace (auto segments)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --no-cpu-baseline --steps 20 > /dev/null 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1) ; [ -n "$f" ] && python -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:8]: print(r['Name'][:58].ljust(58), r['Calls'], r['AverageNs'], r['Percentage'])
"
for seg in 1 13; do
  echo "== SQ counters segments=$seg"
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_$seg -o pmc -- python $REPO/bench.py --no-cpu-baseline --segments $seg --steps 5 --warmup 1 > /dev/null 2> $OUT/sq_$seg.err
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm_$seg -o pmc -- python $REPO/bench.py --no-cpu-baseline --segments $seg --steps 5 --warmup 1 > /dev/null 2> $OUT/grbm_$seg.err
  for d in sq_$seg grbm_$seg; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_interp' not in r.get("Kernel_Name", ""): continue
    agg[r.get("Counter_Name")][0] += float(r.get("Counter_Value", 0)); agg[r.get("Counter_Name")][1] += 1
for c, (v, n) in sorted(agg.items()): print(f"  k_interp {c:22s} per_launch={v/n:.5g}")
PY
  done
done
find $OUT -name "*.csv" -size +2M -delete
echo "== done"
