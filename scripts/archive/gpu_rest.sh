#!/bin/bash
# GPU suite minus the long randomized / large-ensemble / compile-heavy groups (those ran earlier in the round on unchanged kernels)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/rest; mkdir -p $OUT; cd $REPO
timeout ${1:-110} python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 60 -x \
  -k "not randomized and not million and not runtime_models_match and not large_ensemble and not full_size and not lorenz_lsq and not tsit5_cotangent and not offgrid and not checkpointed_fixed_step and not single_rank and not brusselator_lsq and not mlp_matches" \
  --durations=12 > $OUT/pytest.log 2>&1
tail -22 $OUT/pytest.log
