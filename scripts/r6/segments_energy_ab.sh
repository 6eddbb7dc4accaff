#!/bin/bash
# VERDICT r5 next 5, second A/B: the headline pass with 6 time segments (one wave per SIMD) against the planner's choice (12 in 3 groups of 4 waves) and 13 (the plain form),
# on TIME and on board POWER: rocm-smi samples the average socket power every 100 ms while 20000 passes run; energy per pass = mean power x ms per pass.
O=gpurun_out/r6; mkdir -p $O; : > $O/v13_segments_energy.jsonl
for cfg in "0 -" "6 0" "13 0" "24 0"; do
  set -- $cfg; seg=$1; grp=$2
  ( while true; do rocm-smi --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > $O/v13_power_$seg.log &
  SP=$!
  if [ "$grp" = "-" ]; then
    python bench.py --no-extras --no-cpu-baseline --no-pmc --segments $seg --steps 20000 --warmup 100 > $O/v13_bench_$seg.json 2> $O/v13_bench_$seg.err
  else
    HIPADJ_FUSED_GROUP=$grp python bench.py --no-extras --no-cpu-baseline --no-pmc --segments $seg --steps 20000 --warmup 100 > $O/v13_bench_$seg.json 2> $O/v13_bench_$seg.err
  fi
  kill $SP; wait $SP 2>/dev/null
  python - "$seg" "$grp" $O/v13_bench_$seg.json $O/v13_power_$seg.log >> $O/v13_segments_energy.jsonl <<'PY'
import json, re, sys
seg, grp, bj, pl = sys.argv[1:5]
b = json.loads(open(bj).read().strip().splitlines()[-1])
w = []
for line in open(pl):
    m = re.findall(r'"(?:Average Graphics Package Power|Current Socket Graphics Package Power) \(W\)"\s*:\s*"([0-9.]+)"', line)
    if m: w.append(float(m[0]))
w.sort()
hi = [x for x in w if x > 0.7 * (w[-1] if w else 0)]      # the samples taken while the passes ran (the python start-up and the parity leg sit well below)
print(json.dumps(dict(segments_requested=int(seg), group_env=grp, time_segments=b["config"].get("time_segments"), waves_per_workgroup=b["config"].get("waves_per_workgroup"), ms_per_step=b["ms_per_step"], frac=b["roofline"]["frac"],
                      power_samples=len(w), power_W_max=(w[-1] if w else None), power_W_mean_under_load=(sum(hi) / len(hi) if hi else None), millijoule_per_pass=(sum(hi) / len(hi) * b["ms_per_step"] if hi else None))))
PY
done
cat $O/v13_segments_energy.jsonl
head -c 600 $O/v13_power_0.log
