#!/bin/bash
# the configuration the driver runs at round end: torch in the process (its HIP runtime loads first), smoke() then a short bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/torchcheck; mkdir -p $OUT; cd $REPO
( time timeout 140 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
timeout 70 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-700 $OUT/bench.json; tail -2 $OUT/bench.err
