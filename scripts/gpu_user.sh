#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== tsit5/runtime/mixed tests" ; timeout 1500 python -m pytest tests -m gpu -x -q -k "tsit5 or runtime or mixed or native" 2>&1 | grep -v "^  File" | tail -12
echo "== full gpu suite" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
