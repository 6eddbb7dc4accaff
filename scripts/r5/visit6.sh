cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v6
timeout 900 python -m pytest tests/test_gpu_etd.py -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/v6/gpu_tests.log 2>&1
tail -40 gpurun_out/v6/gpu_tests.log | cut -c1-300
HIPADJ_HOST_TIMING=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --loss-paths-only --no-pmc > gpurun_out/v6/bench.json 2> gpurun_out/v6/bench.err
HIPADJ_COT_INPLACE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --loss-paths-only --no-pmc > gpurun_out/v6/bench_launch.json 2> gpurun_out/v6/bench_launch.err
grep "upload_block\|hipadj_adjoint:" gpurun_out/v6/bench.err | tail -12
python - <<'PY'
import json
for f in ('bench', 'bench_launch'):
    d=json.load(open(f'gpurun_out/v6/{f}.json'))
    print(f, d['ms_per_step'], d['cold_burst'], d['roofline']['frac'])
    lp = d.get('loss_paths') or {}
    for k, v in lp.items():
        print(' ', k, {a: b for a, b in v.items() if a != 'note'})
PY
