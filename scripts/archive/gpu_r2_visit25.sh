#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v25; mkdir -p $OUT; cd $REPO
python -c "
import sys; sys.path.insert(0, '.')
from scimlsensitivity_jl_amd import _lib; print(_lib.runtime_compiler())" 2>&1 | tail -1 | tee $OUT/compiler.log
timeout 300 env DBG_TAG=fixed54 DBG_NTRAJ=54 python scripts/repro_hiprtc_compiler.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids | tail -3 | tee -a $OUT/compiler.log
timeout 300 env DBG_TAG=old54 DBG_NTRAJ=54 HIPADJ_HIPRTC=libhiprtc.so python scripts/repro_hiprtc_compiler.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids | tail -3 | tee -a $OUT/compiler.log
timeout 2400 python -m pytest tests/test_gpu_fuzz_mm_events.py tests/test_gpu_mass_matrix.py tests/test_gpu_events.py tests/test_gpu_user_models.py -q -p no:cacheprovider -x 2>&1 | tail -8 | tee $OUT/pytest_rt.log
