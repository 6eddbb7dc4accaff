#!/bin/bash
# GPU visit 22: register stage rows for runtime lane models (adaptive Tsit5): A/B timing, then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v22; O=gpurun_out/r3v22
for v in 1 0 1; do HIPADJ_TS5_REGS_USER=$v timeout 300 python scripts/r3/bench_tsit5_user.py 2>/dev/null | tee -a $O/user_ts5_ab.jsonl; done
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log
