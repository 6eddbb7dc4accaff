#!/bin/bash
# usage: trace.sh <tag> <env assignments...> -- <python script and args>: rocprofv3 --kernel-trace --stats of the command, the kernel stats table to gpurun_out/r4t/<tag>.csv and stdout
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r4t; mkdir -p $OUT
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
env "${envs[@]}" PYTHONWARNINGS=ignore timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$tag -o trace -- python "$@" > $OUT/$tag.out 2> $OUT/$tag.err
find $OUT/tr_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/$tag.csv; rm -rf $OUT/tr_$tag
echo "== $tag"; cat $OUT/$tag.out | tail -3; python - $OUT/$tag.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
