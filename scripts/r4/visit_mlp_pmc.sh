#!/bin/bash
# counters of configs[3]'s kernel (k_mlp_adjoint_grad<128>): how busy the matrix pipe is, what the waves wait for
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r4mlp; mkdir -p $O; rm -f $O/*
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  PROF_ONLY=mlp PROF_REPS=2 timeout 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/r4/prof_families.py > $O/run_$tag.log 2> $O/run_$tag.err
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $O/pmc_mlp.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:48], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    if "mlp_adjoint_grad" in k: print(f"{k:48s} {c:30s} per_launch={v / max(1, n) * 1:.4e} rows={n}")
PY
  tail -2 $O/run_$tag.err | grep -i "error\|invalid\|not" | head -3
done
cd $GRAFT_REPO_ROOT; rm -rf $O/pmc_*/
