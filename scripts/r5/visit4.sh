cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v4
timeout 900 python -m pytest tests/test_gpu_device_loss.py tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_wide.py -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/v4/gpu_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/v4/gpu_tests.log | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/v4/bench.json 2> gpurun_out/v4/bench.err
HIPADJ_COT_INPLACE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > gpurun_out/v4/bench_launch.json 2> gpurun_out/v4/bench_launch.err
python - <<'PY'
import json
for f in ('bench', 'bench_launch'):
    d=json.load(open(f'gpurun_out/v4/{f}.json'))
    print(f, d['ms_per_step'], d['cold_burst'], d['roofline']['frac'])
    lp = d.get('loss_paths') or {}
    for k, v in lp.items():
        print(' ', k, {a: b for a, b in v.items() if a != 'note'})
    for r in d.get('other_configs', []):
        c = r.get('config', '')
        if 'PUBLISHED' in c and '4096' in c: print('  ', c[-30:], 'rev', r.get('reverse_ms'), 'k', r.get('sweep_kernel_ms'))
PY
for v in "" "PROBE_TORCH=1" "PROBE_TSTREAM=1" "PROBE_TDELTA=1" "PROBE_TORCH=1 OMP_NUM_THREADS=1"; do
  env $v VARIANT="registered $v" HIPADJ_HOST_TIMING=1 python scripts/r5/host_probe.py >> gpurun_out/v4/host_probe.jsonl 2>> gpurun_out/v4/host_probe.err
done
cat gpurun_out/v4/host_probe.jsonl
grep upload_block gpurun_out/v4/host_probe.err | awk '{print $11, $14}' | sort | uniq -c | sort -rn | head -12
