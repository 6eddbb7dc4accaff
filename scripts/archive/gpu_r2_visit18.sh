#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v18; mkdir -p $OUT; cd $REPO
timeout 600 python scripts/bench_families.py 2>$OUT/fam.err | tee $OUT/families.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k!='dp'})"
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/scripts/bench_families.py > /dev/null 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1) ; [ -n "$f" ] && python -c "
import csv
for r in list(csv.DictReader(open('$f')))[:16]: print(r['Name'][:84].ljust(84), r['Calls'], r['AverageNs'], r['Percentage'])
"
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
