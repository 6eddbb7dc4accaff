#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
one() { python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('timing','$HIPADJ_TIMING','ms/step %.4f kernel_ms %.4f e2e %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['end_to_end_adjoint_ms']))"; }
for rep in 1 2; do
HIPADJ_TIMING=2 one
HIPADJ_TIMING=1 one
HIPADJ_TIMING=0 one
done
cd /tmp; export TMPDIR=/tmp
HIPADJ_TIMING=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/exp2 -o trace -- python $REPO/bench.py --no-cpu-baseline --steps 40 > /dev/null 2>&1
f=$(find $REPO/gpurun_out/exp2 -name "*kernel_stats.csv" | head -1) ; [ -n "$f" ] && python -c "
import csv
for r in list(csv.DictReader(open('$f')))[:5]: print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
"
f=$(find $REPO/gpurun_out/exp2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# gaps between consecutive kernels in the steady state
names=[r['Kernel_Name'].split('<')[0].split('::')[-1][:20] for r in rows]
st=[int(r['Start_Timestamp']) for r in rows]; en=[int(r['End_Timestamp']) for r in rows]
for i in range(len(rows)-12, len(rows)-1):
    print(names[i].ljust(20), 'dur %.1f us' % ((en[i]-st[i])/1e3), 'gap to next %.1f us' % ((st[i+1]-en[i])/1e3))
PY
rm -rf $REPO/gpurun_out/exp2
