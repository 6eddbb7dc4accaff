#!/bin/bash
# HBM traffic (PMC) of every sensealg's kernels on the Lorenz ensemble: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_algs; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o pmc -- python $REPO/scripts/bench_algs.py 10000 > $OUT/algs_$c.jsonl 2> $OUT/$c.err
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name", "?")
    m = re.match(r"(?:void )?(?:hipadj::)?(\w+)(<[^(]*>)?", name)
    k = (m.group(1) + (m.group(2) or ""))[:70] if m else name[:70]
    agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for k, (v, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print(f"{sys.argv[2]:11s} {k:72s} launches={n:4d} per_launch_KB={v/n:.6g}")
PY
done
find $OUT -name "*.csv" -size +1M -delete
