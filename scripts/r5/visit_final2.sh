# round 5, the evidence visit on the round's last library: the default bench line as the driver runs it, the rocprofv3 kernel summary of the bench command, smoke(), the whole GPU suite
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/vf2
HIPADJ_BENCH_TRACE=1 timeout 500 python bench.py > gpurun_out/vf2/bench.json 2> gpurun_out/vf2/bench.err; echo "bench rc=$?" >> gpurun_out/vf2/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/vf2/bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'), d['cold_burst']['ms_per_step'], d.get('cpu_baseline',{}).get('value'), d.get('secondary_figures_incomplete'))
for k, v in (d.get('loss_paths') or {}).items(): print(' ', k, v.get('ms_per_step', v.get('adjoint_ms')), v.get('over_headline'))
PY
HIPADJ_BENCH_SUPERVISE=0 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/vf2/prof -o r5f2 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > gpurun_out/vf2/prof_bench.json 2> gpurun_out/vf2/prof.err
ls gpurun_out/vf2/prof | head
python - <<'PY'
# ROCm 7 writes a rocpd database: the kernel summary as CSV + the per-phase durations of the headline kernel
import glob, sqlite3, csv
dbs = glob.glob('gpurun_out/vf2/prof/**/*.db', recursive=True)
rows = []
for db in dbs:
    con = sqlite3.connect(db)
    try:
        rows += con.execute("select name, start, end from kernels").fetchall()
    except Exception as e:
        print('db', db, e)
by = {}
for name, s, e in rows: by.setdefault(name, []).append(e - s)
with open('gpurun_out/vf2/kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs'])
    for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])): w.writerow([name, len(d), sum(d), sum(d) / len(d), min(d), max(d)])
main = [(s, e - s) for name, s, e in rows if 'k_interp_fused' in name]
main.sort()
d = [x[1] for x in main]
if d:
    n = len(d)
    with open('gpurun_out/vf2/kernel_phases.txt', 'w') as f:
        f.write(f"k_interp_fused launches: {n}; average over all {sum(d)/n:.0f} ns\n")
        f.write(f"last 20 launches (the timed region of --steps 20): average {sum(d[-20:])/20:.0f} ns, min {min(d[-20:])} max {max(d[-20:])}\n")
        f.write(f"first 25 launches: average {sum(d[:25])/25:.0f} ns; launches 25-75: average {sum(d[25:75])/max(1,len(d[25:75])):.0f} ns\n")
    print(open('gpurun_out/vf2/kernel_phases.txt').read())
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/vf2/smoke.log 2>&1; tail -2 gpurun_out/vf2/smoke.log
# (the three wide-family files ran on this round's unchanged wide kernels in visit 13 — 299 passed, profiles/r5_gpu_suite_wide.log — and are left out here for the GPU-minute budget)
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -rxX --ignore=tests/test_gpu_wide.py --ignore=tests/test_gpu_fuzz_wide.py --ignore=tests/test_gpu_wide_events.py > gpurun_out/vf2/gpu_tests.log 2>&1
tail -6 gpurun_out/vf2/gpu_tests.log | cut -c1-300
