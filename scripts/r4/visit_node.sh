#!/bin/bash
# the unit-local bodies of one-hidden-layer dense chains: the published neural ODE at N = 4096, every wide-family parity test
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r4node; mkdir -p $O; rm -f $O/*
timeout 300 python scripts/r4/node_bench.py > $O/node.log 2> $O/err.log
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_fuzz_wide.py tests/test_gpu_at_size.py -x -q -m gpu > $O/tests.log 2>&1
cat $O/node.log; tail -3 $O/tests.log
