#!/bin/bash
# GPU visit 9: the whole -m gpu suite after the runtime-lane fused cap + the self-test tie-break
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v9; O=gpurun_out/r3v9
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_suite.log 2>&1
tail -15 $O/gpu_suite.log
