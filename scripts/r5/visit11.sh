cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v11
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py tests/test_gpu_etd.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "brusselator or bruss or config5 or config2_device or etd" > gpurun_out/v11/gpu_tests.log 2>&1
tail -25 gpurun_out/v11/gpu_tests.log | cut -c1-300
