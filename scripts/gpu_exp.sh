#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
one() { python bench.py --no-cpu-baseline --segments $1 --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg',d['config']['time_segments'],'kmask','$HIPADJ_EXP_KMASK','ms/step %.4f kernel_ms %.4f GB/s %.0f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved']))"; }
for seg in 1 13; do
  one $seg
  HIPADJ_EXP_KMASK=7 one $seg
  HIPADJ_EXP_KMASK=63 one $seg
done
