#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== pytest runtime models" ; timeout 1500 python -m pytest tests -m gpu -x -q -k "runtime or native" 2>&1 | grep -v "^  File" | tail -25
