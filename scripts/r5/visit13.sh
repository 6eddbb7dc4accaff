#!/bin/bash
# round 5, visit 13: the wide family's last refusals (off-grid x Backsolve / Quadrature, checkpointing on the adaptive solution) and off-grid Quadrature for runtime lane models
mkdir -p gpurun_out/v13
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_fuzz_wide.py tests/test_gpu_wide_events.py -q -m gpu -x --durations=15 -p no:cacheprovider > gpurun_out/v13/wide.log 2>&1
echo "wide rc=$?" >> gpurun_out/v13/wide.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "offgrid" -p no:cacheprovider > gpurun_out/v13/offgrid.log 2>&1
echo "offgrid rc=$?" >> gpurun_out/v13/offgrid.log
tail -5 gpurun_out/v13/wide.log gpurun_out/v13/offgrid.log
