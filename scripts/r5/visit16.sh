#!/bin/bash
# round 5, visit 16: the default bench exactly as the driver runs it (with the progress trace on stderr), then the GaussKronrod x checkpointing case
mkdir -p gpurun_out/v16
export HIPADJ_BENCH_TRACE=1
timeout 500 python bench.py > gpurun_out/v16/bench.json 2> gpurun_out/v16/bench.err
echo "bench rc=$?" >> gpurun_out/v16/bench.err
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "with_checkpointing and gausskronrod" -p no:cacheprovider > gpurun_out/v16/gk.log 2>&1
echo "gk rc=$?" >> gpurun_out/v16/gk.log
tail -n 6 gpurun_out/v16/bench.err; tail -n 3 gpurun_out/v16/gk.log
