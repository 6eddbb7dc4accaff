#!/bin/bash
# GPU visit 21: the wide family after the workgroup-size rule (n / 2): all its parity tests, the sweep with threads = 0, the bench rows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v21; O=gpurun_out/r3v21
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_fuzz_wide.py tests/test_gpu_at_size.py -q -p no:cacheprovider -k "wide or published or random" > $O/wide_tests.log 2>&1; tail -3 $O/wide_tests.log
python scripts/r3/wide_threads_sweep.py 2>&1 | grep "threads    0"
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3v21/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
for r in d.get("other_configs", []):
    c=r["config"]
    if "wide model" in c: print("  ", c[:150], "| fwd", r.get("forward_ms"), "rev", r.get("reverse_ms"), "frac", (r.get("roofline") or {}).get("frac"))
P
