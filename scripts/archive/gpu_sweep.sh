#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
one() { python bench.py --no-cpu-baseline --segments $1 --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg',d['config']['time_segments'],'wtop','$HIPADJ_WTOP','ms/step %.4f kernel_ms %.4f '%(d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do
for w in 3.4 5.0; do for seg in 5 6 7 9 13; do HIPADJ_WTOP=$w one $seg; done; done
done
