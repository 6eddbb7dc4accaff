cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v9
timeout 1200 python -m pytest tests/test_gpu_wide_events.py tests/test_gpu_events.py tests/test_gpu_at_size.py -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/v9/gpu_tests.log 2>&1
tail -30 gpurun_out/v9/gpu_tests.log | cut -c1-300
