#!/bin/bash
# usage: gpurun -- 'bash scripts/gpu_k.sh "<pytest -k expression>"'
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | grep -v "^  File" | tail -20
