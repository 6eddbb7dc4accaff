cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/etd_pmc
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/etd_pmc -o pmc -- python scripts/r5/etd_rows.py > gpurun_out/etd_pmc/run.json 2> gpurun_out/etd_pmc/run.err
find gpurun_out/etd_pmc -name "*counter_collection.csv" | head -2
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/etd_pmc/**/*counter_collection.csv', recursive=True)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0][-60:]
        if 'bruss' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
    out = []
    for k, v in agg.items():
        out.append(k + ' launches %d: ' % cnt[k] + ', '.join(f'{a} {b / max(cnt[k], 1):.4g}' for a, b in sorted(v.items())))
    open('gpurun_out/etd_pmc/summary.txt', 'w').write('\n'.join(out) + '\n'); print('\n'.join(out))
PY
