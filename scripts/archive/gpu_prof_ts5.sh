#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ts5prof; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$tag -o pmc -- python $REPO/scripts/prof_tsit5.py > /dev/null 2> $OUT/$tag.err
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'tsit5' not in r.get("Kernel_Name", ""): continue
    k = (('fwd' if 'forward' in r["Kernel_Name"] else 'adj'), r.get("Counter_Name"))
    agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k:4s} {c:24s} per_launch={v/n:.5g}")
PY
done
find $OUT -name "*.csv" -size +1M -delete
