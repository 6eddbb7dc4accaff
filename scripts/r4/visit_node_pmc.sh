#!/bin/bash
# dynamic instruction counts of the neural-ODE kernels after the unit-local bodies (k_wide_adjoint, k_wide_adjoint_ts5; N = 4096)
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r4nodepmc; mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  PROF_ONLY=node_rk4,node_ts5 PROF_REPS=2 timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/r4/prof_families.py > $O/run_$tag.log 2> $O/run_$tag.err
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $O/pmc_node.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:56], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    if "k_wide_adjoint" in k or "k_wide_forward" in k: print(f"{k:56s} {c:24s} per_launch={v / max(1, n) * 1:.4e} rows={n}")
PY
done
cd $GRAFT_REPO_ROOT; rm -rf $O/pmc_*/; cat $O/run_*.log | cut -c1-250
