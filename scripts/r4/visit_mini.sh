#!/bin/bash
# Round 4: the tests touched after the last full-suite visit + the bench line under rocprofv3 with the per-dispatch trace (phases.py)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r4mini; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
timeout 1500 python -m pytest tests/test_wtrace.py tests/test_gpu_fused.py tests/test_gpu_quad.py tests/test_gpu_at_size.py -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
( timeout 900 python bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench.json 2> $OUT/bench.err )
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv; rm -rf $OUT/trace
cd $REPO; python scripts/r4/phases.py $OUT/kernel_trace.csv 20 5 | tee $OUT/kernel_phases.txt
bash scripts/r4/trace.sh fwd_quad3 HIPADJ_QUAD=1 -- $REPO/scripts/r4/fwd_only.py
python - <<'PY'
import json
for f in ("bench", "bench_under_rocprof"):
    r = json.loads(open(f"gpurun_out/r4mini/{f}.json").read().strip().splitlines()[-1])
    print(f, "ms_per_step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"], "cold", r["cold_burst"]["ms_per_step"], "fwd", r["forward_solve_ms"])
PY
