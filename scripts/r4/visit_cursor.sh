#!/bin/bash
# the quad cursor with the record below requested ahead: Tsit5 sweeps, the wait counters, the quad parity tests
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r4cursor; mkdir -p $O; rm -f $O/*
timeout 200 python scripts/r4/ts5_bench.py > $O/ts5_quad.log 2> $O/err.log
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_WAVES"; do
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
timeout 400 python -m pytest tests/test_gpu_quad.py -x -q -m gpu > $O/tests.log 2>&1
cat $O/ts5_quad.log; tail -3 $O/tests.log
