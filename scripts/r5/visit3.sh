cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v3
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/v3/gpu_tests.log 2>&1
tail -5 gpurun_out/v3/gpu_tests.log
HIPADJ_HOST_TIMING=1 VARIANT=registered python scripts/r5/host_probe.py > gpurun_out/v3/host_probe.jsonl 2> gpurun_out/v3/host_probe.err
HIPADJ_NO_PINNED=1 VARIANT=pageable python scripts/r5/host_probe.py >> gpurun_out/v3/host_probe.jsonl 2>> gpurun_out/v3/host_probe.err
HIPADJ_HOST_COPY_THREADS=1 VARIANT=registered_1thread python scripts/r5/host_probe.py >> gpurun_out/v3/host_probe.jsonl 2>> gpurun_out/v3/host_probe.err
HIPADJ_HOST_COPY_THREADS=16 VARIANT=registered_16threads python scripts/r5/host_probe.py >> gpurun_out/v3/host_probe.jsonl 2>> gpurun_out/v3/host_probe.err
cat gpurun_out/v3/host_probe.jsonl; grep upload_block gpurun_out/v3/host_probe.err | sort | uniq -c | sort -rn | head -8
nproc
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/v3/bench.json 2> gpurun_out/v3/bench.err
HIPADJ_COT_INSWEEP=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/v3/bench_noinsweep.json 2> gpurun_out/v3/bench_noinsweep.err
python - <<'PY'
import json
for f in ('bench', 'bench_noinsweep'):
    d=json.load(open(f'gpurun_out/v3/{f}.json'))
    print(f, d['ms_per_step'], d['cold_burst'], d['roofline']['frac'])
    lp = d.get('loss_paths') or {}
    for k, v in lp.items():
        print(' ', k, {a: b for a, b in v.items() if a != 'note'})
PY
