#!/bin/bash
# round 2, visit 5: new feature tests on the device + the whole GPU suite once + a bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v5; mkdir -p $OUT; cd $REPO
echo "== new tests"
HIPADJ_NO_TORCH=1 timeout 900 python -m pytest tests/test_gpu_checkpoint_lists.py tests/test_julia_seam.py tests/test_trace.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_new.log
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_full.log
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "parity_max_rel_du0_vs_oracle", "parity_max_rel_dp_vs_oracle")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["config"]["time_segments"])
PY
