#!/bin/bash
# round 6, visit 27 (HEAD with Rosenbrock23, the semi-explicit DAE path, ContinuousCallback (every sensealg, saved event states, vector conditions, terminate!) and the pipelined host transfers): the whole GPU suite, smoke(), the default bench line (+ extras) and the rocprofv3 kernel statistics of the same command
O=gpurun_out/r6
mkdir -p $O
( time timeout 1100 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/v27_gpu_suite.log 2>&1
tail -n 6 $O/v27_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/v27_smoke.log 2>&1; tail -n 2 $O/v27_smoke.log
timeout 600 python bench.py > $O/v27_bench.json 2> $O/v27_bench.err; cp bench_extras.json $O/v27_bench_extras.json
wc -c $O/v27_bench.json; cut -c1-2200 $O/v27_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/v27_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/$O/v27_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/v27_bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
find $O/v27_prof -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} $O/v27_rocprofv3_kernel_stats.csv
rm -rf $O/v27_prof
head -n 5 $O/v27_rocprofv3_kernel_stats.csv | cut -c1-250
