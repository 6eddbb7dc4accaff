#!/bin/bash
# GPU visit 8: the fuzz failure of visit 7 (seed 47: 8-state ring, Backsolve, one-launch kernel) — which build is wrong
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v8; O=gpurun_out/r3v8
timeout 900 python scripts/r3/fused_wide_lane_probe.py > $O/probe.log 2> $O/probe.err
timeout 600 python -m pytest tests/test_gpu_fuzz_mm_events.py -q -k "combinations" -p no:cacheprovider > $O/fuzz.log 2>&1
tail -5 $O/fuzz.log; cat $O/probe.log; grep -c disagrees $O/probe.err
