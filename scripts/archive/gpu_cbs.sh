#!/bin/bash
# A/B: k_compose_finish in 256-thread (64 trajectories) vs 64-thread (16 trajectories) workgroups (HIPADJ_CBS=64)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
HIPADJ_CBS=64 timeout 600 python -m pytest tests -m gpu -x -q -k "interpolating and not tsit5 and not randomized" 2>&1 | tail -2
for rep in 1 2 3; do for w in 256 64; do HIPADJ_CBS=$w timeout 300 python bench.py --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CBS=$w', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), 'rest', round(d['ms_per_step']-d['roofline']['kernel_ms'],4))"; done; done
