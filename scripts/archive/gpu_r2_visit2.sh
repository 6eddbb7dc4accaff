#!/bin/bash
# round 2, visit 2: FP64 issue-rate micro-benchmark, kernel variants, per-wave trace dump, SQ counters of the headline kernel
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v2; mkdir -p $OUT; cd $REPO
python - <<'PY'
import numpy as np
rng = np.random.default_rng(20240601)
(np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((10000, 3))).tofile("/tmp/kbench_u0.bin")
PY
{
echo "== fp64 issue"; timeout 60 scripts/kbench_fp64
echo "== variants (PF x waves/SIMD x segments)"
for r in 1 2; do
timeout 60 scripts/kbench_ops 10000 13 60 bench 2.7 | tail -1
timeout 60 scripts/kbench_ops_pf4 10000 13 60 bench 2.7 | tail -1
timeout 60 scripts/kbench_ops_pf4w3 10000 19 60 bench 2.7 | tail -1
timeout 60 scripts/kbench_ops_pf3w3 10000 19 60 bench 2.7 | tail -1
timeout 60 scripts/kbench_ops_pf4w3 10000 16 60 bench 2.7 | tail -1
done
echo "== trace dump"
KB_TRACE_DUMP=$OUT/trace_ops_13.txt timeout 60 scripts/kbench_ops 10000 13 1 trace 2.7
KB_TRACE_DUMP=$OUT/trace_ops_pf4w3_19.txt timeout 60 scripts/kbench_ops_pf4w3 10000 19 1 trace 2.7
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -c . $OUT/counters_avail.txt
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -o pmc -- $REPO/scripts/kbench_ops 10000 13 10 bench 2.7 > /dev/null 2> $OUT/pmc_$tag.err
  f=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r.get("Kernel_Name", "?")[:40], r.get("Counter_Name"))
    agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if "k_interp" in k: print(f"{k:40s} {c:28s} per_launch={v/n:.5g}")
PY
done 2>&1 | tee $OUT/pmc_sq.txt
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
