#!/bin/bash
# round 2, first GPU visit: kernel experiments with scripts/kbench (no torch), the at-size parity tests, the new bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v1; mkdir -p $OUT; cd $REPO
python - <<'PY'
import numpy as np
rng = np.random.default_rng(20240601)
u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((10000, 3))
u0.tofile("/tmp/kbench_u0.bin")
PY
{
echo "== A/B generic vs stage-operator step (auto plan)"
for b in noops ops noops ops; do timeout 60 scripts/kbench_$b 10000 0 60 bench | tail -1; done
echo "== w_top sweep (ops)"
for w in 2.4 2.7 3.0 3.4 3.8; do echo "wtop $w"; timeout 60 scripts/kbench_ops 10000 0 60 bench $w | tail -1; done
echo "== segment sweep"
for c in 5 6 7 9 11 13 16 19; do timeout 60 scripts/kbench_ops 10000 $c 60 bench | tail -1; done
echo "== segment sweep (generic)"
for c in 6 13; do timeout 60 scripts/kbench_noops 10000 $c 60 bench | tail -1; done
echo "== shard sizes"
for n in 1250 2500 5000; do timeout 60 scripts/kbench_ops $n 0 60 bench | tail -2; done
echo "== wave trace"
timeout 60 scripts/kbench_noops 10000 0 1 trace
timeout 60 scripts/kbench_ops 10000 0 1 trace
timeout 60 scripts/kbench_ops 10000 6 1 trace
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
echo "== pytest at size + quick subset"
timeout 900 python -m pytest tests/test_gpu_at_size.py tests/test_gpu_parity.py::test_torch_autograd_device_path -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_at_size.log
HIPADJ_NO_TORCH=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "lorenz_lsq or segmentation or full_size or golden_gradient_lorenz or repeated_calls" 2>&1 | tail -5 | tee $OUT/pytest_subset.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cat $OUT/bench.json
HIPADJ_NO_OPS=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_noops.json 2>/dev/null; cat $OUT/bench_noops.json
