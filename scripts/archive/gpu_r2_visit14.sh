#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v14; mkdir -p $OUT; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -k "mlp or config4 or C4" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_mlp.log
timeout 600 python scripts/bench_families.py 2>$OUT/fam.err | head -4 | tee $OUT/families.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k!='dp'})"
