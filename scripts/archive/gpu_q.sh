#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests -m gpu -x -q -k "tsit5 and (quadrature or all_models or golden)" 2>&1 | tail -2
timeout 300 python scripts/bench_tsit5_quad.py 2>/dev/null
