cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v1
python -m pytest tests/test_gpu_device_loss.py tests/test_julia_seam.py -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/v1/new_tests.log 2>&1
tail -40 gpurun_out/v1/new_tests.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "single_rank_rccl or lorenz_lsq_matches_oracle or runtime_lv_equals" > gpurun_out/v1/old_subset.log 2>&1
tail -5 gpurun_out/v1/old_subset.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/v1/bench.json 2> gpurun_out/v1/bench.err
tail -c 3000 gpurun_out/v1/bench.json; tail -5 gpurun_out/v1/bench.err
