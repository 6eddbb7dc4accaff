cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/vf
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider -rxX > gpurun_out/vf/gpu_tests.log 2>&1
tail -8 gpurun_out/vf/gpu_tests.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/vf/smoke.log 2>&1; tail -3 gpurun_out/vf/smoke.log
python bench.py > gpurun_out/vf/bench.json 2> gpurun_out/vf/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/vf/bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cold_burst']['ms_per_step'], d.get('cpu_baseline',{}).get('value'))
for k, v in (d.get('loss_paths') or {}).items(): print(' ', k, v.get('ms_per_step', v.get('adjoint_ms')), v.get('over_headline'))
for r in d.get('other_configs', []):
    c = r.get('config', '')
    if 'exponential' in c or 'dense chains' in c: print('  ', c[:110], {k: v for k, v in r.items() if k.endswith('_ms') or k in ('error',)}, [(x['H'], round(x['mfma_speedup_reverse'],1)) for x in r.get('dense_chain_crossover', [])])
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/vf/prof -o r5f -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > gpurun_out/vf/prof_bench.json 2> gpurun_out/vf/prof.err
ls gpurun_out/vf/prof
