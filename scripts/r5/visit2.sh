cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v2
python -m pytest tests/test_gpu_device_loss.py -m gpu -q --maxfail=30 -p no:cacheprovider -k "lsq_data or streaming or virtual_shards" > gpurun_out/v2/new_tests.log 2>&1
tail -3 gpurun_out/v2/new_tests.log
python scripts/r5/ab_probe.py nz1 > gpurun_out/v2/nz1.jsonl 2>&1; cat gpurun_out/v2/nz1.jsonl
python scripts/r5/ab_probe.py ticket > gpurun_out/v2/ticket.jsonl 2>&1; cat gpurun_out/v2/ticket.jsonl
python scripts/r5/shard_study.py 1250 10000 > gpurun_out/v2/shard_study.jsonl 2>&1; cat gpurun_out/v2/shard_study.jsonl
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/v2/bench.json 2> gpurun_out/v2/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/v2/bench.json'))
print(d['ms_per_step'], d['cold_burst'], d['roofline']['frac'])
print(json.dumps(d.get('loss_paths'), indent=1))
PY
