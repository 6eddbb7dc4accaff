#!/bin/bash
# GPU visit 12: adaptive Tsit5 in the wide family (forward + GaussAdjoint), first run
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v12; O=gpurun_out/r3v12
timeout 900 python -m pytest tests/test_gpu_wide.py -q -p no:cacheprovider -x > $O/wide_ts5.log 2>&1
tail -40 $O/wide_ts5.log
