#!/bin/bash
# Round 3, last GPU minutes: the -m gpu suite on the round's final commit (the wide-family files ran a moment ago in full: deselected here to fit the budget)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v25; O=gpurun_out/r3v25
timeout 560 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_wide.py --deselect tests/test_gpu_fuzz_wide.py > $O/gpu_suite_rest.log 2>&1; tail -6 $O/gpu_suite_rest.log
