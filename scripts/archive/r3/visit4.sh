#!/bin/bash
# Round 3, GPU visit 4: fused Gauss / Backsolve, Tsit5 (trimmed stage sums, record start), LDS exchange floor, wide rows inside the whole bench line.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v4; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 900 python -m pytest tests/test_gpu_fused.py -q -p no:cacheprovider -x 2>&1 | tail -8 ) | tee $OUT/fused_tests.log
( timeout 1200 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "tsit5 or Tsit5 or adaptive" 2>&1 | tail -8 ) | tee $OUT/tsit5_tests.log
scripts/r3/lds_exchange_floor 2>&1 | tee $OUT/lds_floor.log
timeout 600 python scripts/bench_tsit5.py 10000 2>/dev/null | tee $OUT/tsit5.jsonl | cut -c1-260
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3 | tee $OUT/bench_time.log
python - <<'PY'
import json
r = json.load(open("gpurun_out/r3v4/bench.json"))
print("ms_per_step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "region", r["roofline"]["region_event_ms_per_step"], "fwd", r["forward_solve_ms"])
for s in r.get("shard_sizes", []): print(" shard", s["ntraj"], s["ms_per_step"], s["kernel_ms"])
for c in r.get("other_configs", []): print(" ", c.get("config", "")[:110], "| rev", c.get("reverse_ms"), "| kms", c.get("sweep_kernel_ms", c.get("main_kernel_ms")), "| frac", (c.get("roofline") or {}).get("frac"), c.get("error", ""))
print(r.get("other_configs_error"))
PY
