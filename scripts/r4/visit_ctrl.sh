#!/bin/bash
# the cheaper step-size controller (one log + one exp, inlined) and the quad cursor's look-ahead: Tsit5 sweeps of the quad, lane and wide families, every adaptive parity test
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r4ctrl; mkdir -p $O; rm -f $O/*
timeout 200 python scripts/r4/ts5_bench.py > $O/ts5_quad.log 2> $O/err.log
HIPADJ_QUAD=0 TS5_TOLS=default timeout 200 python scripts/r4/ts5_bench.py > $O/ts5_lane.log 2>> $O/err.log
timeout 300 python scripts/r4/node_bench.py > $O/node.log 2>> $O/err.log
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_gpu_wide.py tests/test_gpu_parity.py tests/test_gpu_checkpoint_lists.py tests/test_gpu_events.py tests/test_gpu_mass_matrix.py tests/test_gpu_fuzz_wide.py -x -q -m gpu -k "tsit5 or ts5 or adaptive or Tsit5 or quad or fuzz" > $O/tests.log 2>&1
cat $O/ts5_quad.log $O/ts5_lane.log $O/node.log; tail -3 $O/tests.log
