#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/ts5
timeout 600 python scripts/bench_tsit5.py 10000 2>&1 | tee gpurun_out/ts5/tsit5.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['model'], d['alg'], d['abstol'], 'fwd %.3f adj %.3f ms'%(d['forward_ms'], d['adjoint_ms']))"
