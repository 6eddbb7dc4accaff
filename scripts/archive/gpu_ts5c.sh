#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/ts5
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -x -q -k "tsit5 or runtime_models or gausskronrod or mixed or randomized" 2>&1 | tail -4
timeout 600 python scripts/bench_tsit5.py 10000 2>&1 | tee gpurun_out/ts5/tsit5.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d.get('model'), d.get('alg'), d.get('abstol'), 'fwd %.3f adj %.3f ms ws %.3f GB'%(d.get('forward_ms',0), d.get('adjoint_ms',0), d.get('workspace_GB',0)), d.get('error',''))"
