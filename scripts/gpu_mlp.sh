#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "mlp" 2>&1 | tail -5
timeout 900 python scripts/bench_families.py 2>/dev/null | head -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k!='dp'})"
