#!/bin/bash
# A/B: k_interp in single-wave workgroups vs 256-thread workgroups (HIPADJ_WPB=4); parity subset first
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
HIPADJ_WPB=4 timeout 600 python -m pytest tests -m gpu -x -q -k "interpolating and not tsit5 and not randomized" 2>&1 | tail -3
for rep in 1 2 3; do for w in 1 4; do HIPADJ_WPB=$w timeout 300 python bench.py --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WPB=$w', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], round(d['roofline']['frac'],3))"; done; done
