#!/bin/bash
# GPU visit 10: adaptive Tsit5 with the stage rows in registers (compiled-in models) vs the LDS rows (build_ab/libhipadj_ts5lds.so, -DHIPADJ_TS5_REGS=0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v10; O=gpurun_out/r3v10
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "tsit5 or Tsit5 or adaptive or checkpoint_lists or events or mass_matrix" > $O/tsit5_tests.log 2>&1
tail -4 $O/tsit5_tests.log
echo "== registers (shipped)"; timeout 600 python scripts/bench_tsit5.py > $O/tsit5_regs.jsonl 2> $O/tsit5_regs.err
echo "== LDS rows"; HIPADJ_LIBRARY=$PWD/build_ab/libhipadj_ts5lds.so timeout 600 python scripts/bench_tsit5.py > $O/tsit5_lds.jsonl 2> $O/tsit5_lds.err
echo "== registers again"; timeout 600 python scripts/bench_tsit5.py > $O/tsit5_regs2.jsonl 2>> $O/tsit5_regs.err
python - <<'P'
import json
for f in ("tsit5_regs","tsit5_lds","tsit5_regs2"):
    for l in open(f"gpurun_out/r3v10/{f}.jsonl"):
        d=json.loads(l)
        if "error" in d: print(f, d); continue
        print(f"{f:12s} {d['model']:7s} {d['alg']:14s} tol {d['abstol']:.0e}/{d['reltol']:.0e}  fwd {d['forward_ms']:.3f}  rev kernel {d['adjoint_kernel_ms']:.3f}  dp0 {d['dp'][0]:.12e}")
P
