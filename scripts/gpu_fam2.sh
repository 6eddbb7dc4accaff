#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "mlp" 2>&1 | tail -3
bash scripts/gpu_families.sh ${1:-fam2}
