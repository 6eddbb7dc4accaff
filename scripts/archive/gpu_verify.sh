#!/bin/bash
# regression subset, the randomized test with fresh seeds, then the whole GPU suite. usage: gpurun -- 'bash scripts/gpu_verify.sh <fuzz bases...>'
cd ${GRAFT_REPO_ROOT:-.}
echo "== no-loss-times regression"; timeout 600 python -m pytest tests -m gpu -x -q -k "without_loss_times" 2>&1 | grep -v "^  File" | tail -6
for base in "$@"; do echo "== fuzz base $base"; HIPADJ_FUZZ_BASE=$base timeout 900 python -m pytest tests -m gpu -q -k "randomized_configurations_match" 2>&1 | grep -v "^  File" | tail -6; done
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
