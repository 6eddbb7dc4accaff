#!/bin/bash
# GPU visit 18: adaptive checkpointed sweeps with register rows (shipped) vs LDS rows (build_ab/libhipadj_ckld.so), and their parity tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v18; O=gpurun_out/r3v18
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "checkpoint or ckpt" > $O/ckpt_tests.log 2>&1; tail -3 $O/ckpt_tests.log
echo "== registers"; timeout 300 python scripts/r3/bench_tsit5_ckpt.py | tee $O/regs.jsonl
echo "== LDS"; HIPADJ_LIBRARY=$PWD/build_ab/libhipadj_ckld.so timeout 300 python scripts/r3/bench_tsit5_ckpt.py | tee $O/lds.jsonl
echo "== registers"; timeout 300 python scripts/r3/bench_tsit5_ckpt.py | tee $O/regs2.jsonl
