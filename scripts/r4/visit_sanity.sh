#!/bin/bash
# last code of the round: the families touched after the full-suite visit (wide, quad, traced, at size), smoke, the bench line
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r4sanity; mkdir -p $O; rm -f $O/*
export PYTHONWARNINGS=ignore
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_quad.py tests/test_wtrace.py tests/test_gpu_at_size.py tests/test_gpu_fuzz_wide.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); tail -c 200 $O/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4sanity/bench.json").read().strip().splitlines()[-1])
rf = r["roofline"]
print("ms_per_step", r["ms_per_step"], "value", r["value"], "kernel_ms", rf["kernel_ms"], "frac", rf["frac"], "traffic", rf["traffic"], "cold", r["cold_burst"]["ms_per_step"], "fwd", r["forward_solve_ms"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
for o in r.get("other_configs", []):
    if "neural" in o["config"] and "4096" in o["config"]: print(o["config"][-60:], o.get("forward_ms"), o["reverse_ms"])
PY
