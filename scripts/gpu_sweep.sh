#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
one() { python bench.py --no-cpu-baseline --segments $1 --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg',d['config']['time_segments'],'wtop','$HIPADJ_WTOP','traj/s %.3e ms/step %.4f kernel_ms %.4f GB/s %.0f'%(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved']))"; }
for rep in 1 2; do
for w in 3.4 4.5 5.5 6.5 8.0; do for seg in 12 13; do HIPADJ_WTOP=$w one $seg; done; done
done
