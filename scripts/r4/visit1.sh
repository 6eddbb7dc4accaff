#!/bin/bash
# Round 4 visit 1: the bench line with the self-consistent roofline + live PMC traffic; rocprofv3 kernel stats of the same command; kernel stats + FETCH / WRITE
# PMC for the secondary kernels (k_wide_adjoint, k_wide_adjoint_ts5, k_mlp_adjoint_grad<128>, k_bruss_quad_adj<32>); SQ / GRBM counters of k_interp_fused; sclk + power trace
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r4v1; mkdir -p $OUT; cd $REPO
export PYTHONWARNINGS=ignore
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ); tail -c 300 $OUT/bench.err
timeout 300 python scripts/r4/clock_trace.py > $OUT/clock_trace.json 2> $OUT/clock_trace.err; tail -c 300 $OUT/clock_trace.err
ls /sys/class/drm/ > $OUT/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/ >> $OUT/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/* >> $OUT/sysfs_ls.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; rm -rf $OUT/trace
# SQ / GRBM counters of the headline kernel (one pass: 7 SQ + 1 GRBM)
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o sq -- python $REPO/bench.py --pmc-child > /dev/null 2> $OUT/sq.err
f=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/sq_summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_interp" in r["Kernel_Name"] or "k_forward" in r["Kernel_Name"]: acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()): print(k, c, "launches", len(v), "mean", sum(v) / len(v), "min", min(v), "max", max(v))
PY
rm -rf $OUT/sq
# the secondary kernels
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ftrace -o trace -- python $REPO/scripts/r4/prof_families.py > $OUT/families_cases.jsonl 2> $OUT/ftrace.err
find $OUT/ftrace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/families_kernel_stats.csv
find $OUT/ftrace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/families_kernel_trace.csv; rm -rf $OUT/ftrace
for c in FETCH_SIZE WRITE_SIZE; do
  PROF_REPS=2 timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/fpmc_$c -o pmc -- python $REPO/scripts/r4/prof_families.py > /dev/null 2> $OUT/fpmc_$c.err
  find $OUT/fpmc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/families_pmc_$c.csv; rm -rf $OUT/fpmc_$c
done
cd $REPO
python scripts/r4/join_prof.py $OUT/families_cases.jsonl $OUT/families_kernel_trace.csv $OUT/families_pmc_FETCH_SIZE.csv $OUT/families_pmc_WRITE_SIZE.csv | tee $OUT/families_roofline.jsonl
python - <<'PY'
import json, csv
r = json.loads(open("gpurun_out/r4v1/bench.json").read().strip().splitlines()[-1])
rf = r["roofline"]
print("ms_per_step", r["ms_per_step"], "kernel_ms", rf["kernel_ms"], "frac", rf["frac"], "whole", rf["whole_pass_frac"], "dispatch_ev", rf["dispatch_event_kernel_ms"], "traffic", rf["traffic"], rf["traffic_source"][:80], "fwd", r["forward_solve_ms"])
for s in r.get("shard_sizes", []): print(" shard", s["ntraj"], s["ms_per_step"], s["kernel_ms"], s["implied_speedup_if_allreduce_hidden"])
for o in r.get("other_configs", []): print("  ", o["config"][:110], "| rev", o.get("reverse_ms"), "| frac", (o.get("roofline") or {}).get("frac"))
for row in list(csv.reader(open("gpurun_out/r4v1/kernel_stats.csv")))[:6]: print(row[0][:60], row[1:7])
c = json.loads(open("gpurun_out/r4v1/clock_trace.json").read().strip().splitlines()[-1])
print(json.dumps(c["phases"], indent=0)[:3000])
PY
