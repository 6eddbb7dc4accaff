#!/usr/bin/env python
"""phases.py <kernel_trace.csv> [steps warmup]: per-phase averages of the dominant kernel's durations in a `rocprofv3 --kernel-trace` run of
`bench.py --no-extras --no-pmc --no-cpu-baseline --steps K --warmup W`.  The launches of one run, in order: 400 power-preamble passes, W warm-up, K TIMED, 3 + max(10, K) with
dispatch-packet events (r.profiled), then — after 0.3 s of idle — W warm-up and K timed passes of the cold burst.  The summary average of --stats mixes all of them; the timed
region is what bench.py's `kernel_ms` measures."""
import csv
import sys

path = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = [r for r in csv.DictReader(open(path)) if "k_interp" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows]          # us
starts = [int(r["Start_Timestamp"]) for r in rows]
P = len(d) - (W + K) - (3 + max(10, min(K, 50))) - (W + K)                                    # preamble passes
phases = [("power preamble (first 25)", 0, min(25, P)), ("power preamble (passes 25-75: the limiter's clamp)", 25, min(75, P)), ("power preamble (last 100)", max(P - 100, 0), P),
          ("warm-up", P, P + W), ("TIMED REGION", P + W, P + W + K), ("dispatch-event loop", P + W + K, P + W + K + 3 + max(10, min(K, 50))),
          ("cold burst: warm-up", len(d) - (W + K), len(d) - K), ("cold burst: timed", len(d) - K, len(d))]
print(f"{len(d)} launches of {rows[0]['Kernel_Name'][:60]}...; all: mean {sum(d) / len(d):.1f} us, min {min(d):.1f}, max {max(d):.1f}")
for name, a, b in phases:
    if b > a:
        seg = d[a:b]
        span = (int(rows[b - 1]["End_Timestamp"]) - starts[a]) * 1e-3 / (b - a)
        print(f"  {name:52s} launches {a:3d}..{b - 1:3d}: kernel mean {sum(seg) / len(seg):7.1f} us (min {min(seg):6.1f}, max {max(seg):6.1f}); start-to-end span per launch {span:7.1f} us")
