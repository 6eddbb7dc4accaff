#!/bin/bash
# GPU visit 19: forward-record prefetch in the adaptive reverse sweeps (shipped) vs without (build_ab/libhipadj_nopf.so) + the adaptive parity tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v19; O=gpurun_out/r3v19
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "tsit5 or Tsit5 or adaptive" > $O/ts5_tests.log 2>&1; tail -3 $O/ts5_tests.log
for tag in pf nopf pf2; do
  lib=""; [ $tag = nopf ] && lib=$PWD/build_ab/libhipadj_nopf.so
  HIPADJ_LIBRARY=$lib timeout 600 python scripts/bench_tsit5.py > $O/$tag.jsonl 2>/dev/null
done
python - <<'P'
import json
for f in ("pf","nopf","pf2"):
    for l in open(f"gpurun_out/r3v19/{f}.jsonl"):
        d=json.loads(l)
        if "error" in d: print(f, d); continue
        print(f"{f:5s} {d['model']:7s} {d['alg']:14s} tol {d['abstol']:.0e}/{d['reltol']:.0e}  fwd {d['forward_ms']:.3f}  rev kernel {d['adjoint_kernel_ms']:.3f}  dp0 {d['dp'][0]:.12e}")
P
