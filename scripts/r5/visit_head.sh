# round 5, last visit: the default bench line, smoke() and the wide-family test files on the final HEAD (the evidence visit scripts/r5/visit_final2.sh ran the rest of the suite
# on the library before GaussKronrod over the reverse step list was added to the wide family)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/vh
timeout 200 python bench.py > gpurun_out/vh/bench.json 2> gpurun_out/vh/bench.err; echo "bench rc=$?" >> gpurun_out/vh/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/vh/bench.json'))
print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'), d['cold_burst']['ms_per_step'], d.get('secondary_figures_incomplete'), len(d.get('other_configs', [])))
PY
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/vh/smoke.log 2>&1; tail -n 1 gpurun_out/vh/smoke.log
timeout 60 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "offgrid or errors_mirror" -p no:cacheprovider > gpurun_out/vh/parity_offgrid.log 2>&1; tail -n 1 gpurun_out/vh/parity_offgrid.log
timeout 170 python -m pytest tests/test_gpu_wide.py tests/test_gpu_fuzz_wide.py tests/test_gpu_wide_events.py -q -m gpu -p no:cacheprovider > gpurun_out/vh/wide.log 2>&1; tail -n 1 gpurun_out/vh/wide.log
