#!/bin/bash
# round 5, visit 15: where does the default bench die (visit 14: "Write access to a read-only page"), and the GaussKronrod x checkpointing case over the reverse step list
mkdir -p gpurun_out/v15
export HIPADJ_BENCH_TRACE=1
timeout 400 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > gpurun_out/v15/bench.json 2> gpurun_out/v15/bench.err
echo "bench rc=$?" >> gpurun_out/v15/bench.err
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "with_checkpointing and lorenz and default and gausskronrod" -p no:cacheprovider > gpurun_out/v15/gk.log 2>&1
echo "gk rc=$?" >> gpurun_out/v15/gk.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "with_checkpointing and not gausskronrod or stride_or_list or shortened" -p no:cacheprovider > gpurun_out/v15/rest.log 2>&1
echo "rest rc=$?" >> gpurun_out/v15/rest.log
tail -n 12 gpurun_out/v15/bench.err; tail -n 6 gpurun_out/v15/gk.log; tail -n 4 gpurun_out/v15/rest.log
