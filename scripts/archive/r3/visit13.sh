#!/bin/bash
# GPU visit 13: smoke + bench (new adaptive wide rows) on the round's code
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v13; O=gpurun_out/r3v13
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3v13/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "fwd", d["forward_solve_ms"])
for r in d.get("other_configs", []):
    c=r["config"]
    if "wide model" in c: print("  ", c[:150], "| fwd", r.get("forward_ms"), "rev", r.get("reverse_ms"), "grad", r.get("gradient_ms"), "pub", r.get("reference_published_cpu_ms"))
P
