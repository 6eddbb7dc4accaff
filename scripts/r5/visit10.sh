cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v10
timeout 2000 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider -rxX > gpurun_out/v10/gpu_tests.log 2>&1
tail -12 gpurun_out/v10/gpu_tests.log | cut -c1-300
python scripts/r5/host_probe.py > gpurun_out/v10/host_probe.jsonl 2>/dev/null; cat gpurun_out/v10/host_probe.jsonl
