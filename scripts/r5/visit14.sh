#!/bin/bash
# round 5, visit 14: the lane family's off-grid additions (GaussKronrod, checkpointing = true, Backsolve stride / list), the end-of-span slope fix, and the forward kernels' timing after it
mkdir -p gpurun_out/v14
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "offgrid or off_grid or shortened or not_a_multiple or runtime" -p no:cacheprovider --durations=8 > gpurun_out/v14/lane.log 2>&1
echo "lane rc=$?" >> gpurun_out/v14/lane.log
timeout 300 python -m pytest tests/test_gpu_quad.py tests/test_gpu_fused.py tests/test_gpu_at_size.py -q -m gpu -x -p no:cacheprovider > gpurun_out/v14/fwd.log 2>&1
echo "fwd rc=$?" >> gpurun_out/v14/fwd.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/v14/bench.json 2> gpurun_out/v14/bench.err
echo "bench rc=$?" >> gpurun_out/v14/bench.err
tail -n 4 gpurun_out/v14/lane.log; tail -n 3 gpurun_out/v14/fwd.log; tail -n 2 gpurun_out/v14/bench.err
