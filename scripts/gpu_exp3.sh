#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
python -m pytest tests -m gpu -x -q -k "lorenz_lsq or reproducible or segmentation" 2>&1 | tail -2
for ff in 0 1; do
export HIPADJ_FUSED_FINAL=$ff
python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_final','$ff','ms/step %.4f kernel_ms %.4f e2e %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['end_to_end_adjoint_ms']))"
cd /tmp; export TMPDIR=/tmp
HIPADJ_TIMING=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/exp3 -o trace -- python $REPO/bench.py --no-cpu-baseline --steps 40 > /dev/null 2>&1
f=$(find $REPO/gpurun_out/exp3 -name "*kernel_stats.csv" | head -1) ; [ -n "$f" ] && python -c "
import csv
for r in list(csv.DictReader(open('$f')))[:4]: print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
"
rm -rf $REPO/gpurun_out/exp3
cd $REPO
done
