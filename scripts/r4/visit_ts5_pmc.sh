#!/bin/bash
# counters of the quad Tsit5 kernels (Lorenz, 10^4, Interpolating, default tolerances): dynamic instruction counts per wavefront
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4ts5pmc; O=$PWD/gpurun_out/r4ts5pmc
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/r3/tsit5_one.py interpolating > $O/run_$tag.log 2> $O/run_$tag.err
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $O/pmc_tsit5_quad.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:64], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    if "tsit5" in k: print(f"{k:64s} {c:28s} per_launch={v / max(1, n) * 1:.4e} rows={n}")
PY
done
cd $GRAFT_REPO_ROOT; rm -rf $O/pmc_*/
python scripts/r3/tsit5_one.py interpolating > $O/plain.log 2>&1; tail -1 $O/plain.log
HIPADJ_QUAD=0 python scripts/r3/tsit5_one.py interpolating > $O/plain_lane.log 2>&1; tail -1 $O/plain_lane.log
