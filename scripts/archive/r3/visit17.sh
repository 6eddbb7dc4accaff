#!/bin/bash
# GPU visit 17: randomised differential test of the wide family
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v17; O=gpurun_out/r3v17
timeout 900 python -m pytest tests/test_gpu_fuzz_wide.py -q -p no:cacheprovider > $O/fuzz_wide.log 2>&1
tail -60 $O/fuzz_wide.log | cut -c1-330
