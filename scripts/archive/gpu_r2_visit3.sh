#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v3; mkdir -p $OUT; cd $REPO
python - <<'PY'
import numpy as np
rng = np.random.default_rng(20240601)
(np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((10000, 3))).tofile("/tmp/kbench_u0.bin")
PY
{
for r in 1 2; do
for b in ops ops_prio ops_pf4_prio lay_pf4 lay_pf4_prio lay_prio; do echo "-- $b"; timeout 60 scripts/kbench_$b 10000 13 60 bench 2.6 | tail -2; done
done
echo "== 200 reps"; timeout 60 scripts/kbench_ops_prio 10000 13 200 bench 2.6 | tail -2
echo "== traces"
for b in ops_prio lay_pf4_prio; do KB_TRACE_DUMP=$OUT/trace_$b.txt timeout 60 scripts/kbench_$b 10000 13 1 trace 2.6; done
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
