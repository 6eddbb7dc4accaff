cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v7
timeout 600 python -m pytest tests/test_gpu_canary.py -m gpu -q -rxX -p no:cacheprovider > gpurun_out/v7/canary.log 2>&1
tail -6 gpurun_out/v7/canary.log | cut -c1-300
HIPADJ_HOST_TIMING=1 python bench.py --steps 20 --warmup 5 > gpurun_out/v7/bench.json 2> gpurun_out/v7/bench.err
grep "upload_block\|hipadj_adjoint:" gpurun_out/v7/bench.err | tail -8
python - <<'PY'
import json
d=json.load(open('gpurun_out/v7/bench.json'))
print(d['ms_per_step'], d['cold_burst'], d['roofline']['frac'], d.get('cpu_baseline'))
for k, v in (d.get('loss_paths') or {}).items():
    print(' ', k, {a: b for a, b in v.items() if a != 'note'})
for r in d.get('other_configs', []):
    c = r.get('config', '')
    if 'exponential' in c or 'horizon' in c or ('PUBLISHED' in c and '4096' in c): print('  ', c[:150], {k: v for k, v in r.items() if k.endswith('_ms') or k in ('us_per_step', 'workspace_GB', 'error')})
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/v7/prof -o r5 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > gpurun_out/v7/prof_bench.json 2> gpurun_out/v7/prof.err
ls gpurun_out/v7/prof | head; find gpurun_out/v7/prof -name "*stats*" | head
