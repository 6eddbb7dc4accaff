#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "^  File" | tail -22
