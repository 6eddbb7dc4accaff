#!/bin/bash
# shortest possible GPU visit: one test per translation unit of libhipadj.so + the RCCL communicator, WITHOUT torch in the
# process (HIPADJ_NO_TORCH=1: HIP runtime, hiprtc and RCCL of the ROCm installation only; no 1-2 min torch import)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/quick; mkdir -p $OUT; cd $REPO
export HIPADJ_NO_TORCH=1
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 90 \
  -k "${HIPADJ_QUICK_K:-native_library or lorenz_lsq or golden_gradient_lorenz or brusselator_lsq or mlp_matches or tsit5_cotangent or runtime_lv_equals or single_rank}" \
  > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
python - > $OUT/modules.txt 2>&1 <<'PY'
import sys
print("torch" in sys.modules)
PY
