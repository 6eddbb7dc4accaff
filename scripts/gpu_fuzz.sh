#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests -m gpu -x -q -k "cost_function or attached" 2>&1 | grep -v "^  File" | tail -20
