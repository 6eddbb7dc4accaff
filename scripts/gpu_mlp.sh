#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== mlp tests"; timeout 600 python -m pytest tests -m gpu -x -q -k "mlp" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/scripts/bench_families.py > /tmp/fam.out 2> /tmp/prof.err
grep -i mlp /tmp/fam.out | cut -c1-400
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1) ; [ -n "$f" ] && python -c "
import csv
for r in list(csv.DictReader(open('$f')))[:10]: print(r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['Percentage'])
"
