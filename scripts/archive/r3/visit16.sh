#!/bin/bash
# GPU visit 16: continuous costs on wide models (first run) + the at-size adaptive wide tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v16; O=gpurun_out/r3v16
timeout 900 python -m pytest tests/test_gpu_wide.py -q -p no:cacheprovider -k "cost" > $O/wide_cost.log 2>&1
tail -30 $O/wide_cost.log | cut -c1-250
