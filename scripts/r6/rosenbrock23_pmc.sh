#!/bin/bash
# SQ counters of the Rosenbrock23 kernels (forward lane kernel and the reverse kernel of a runtime model), collected alone (no trace domains next to --pmc), two passes
# so that no pass asks for more counters than the SQ has slots.  Workload: scripts/r6/bench_rosenbrock23.py at N = 8192.
#   gpurun --timeout 900 -- 'bash scripts/r6/rosenbrock23_pmc.sh'   ->  gpurun_out/r6/ros23_pmc_summary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6/ros23_pmc_a gpurun_out/r6/ros23_pmc_b
timeout 420 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/r6/ros23_pmc_a -o pmc -- python scripts/r6/bench_rosenbrock23.py 8192 > gpurun_out/r6/ros23_pmc_a/run.json 2> gpurun_out/r6/ros23_pmc_a/run.err
timeout 420 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/r6/ros23_pmc_b -o pmc -- python scripts/r6/bench_rosenbrock23.py 8192 > gpurun_out/r6/ros23_pmc_b/run.json 2> gpurun_out/r6/ros23_pmc_b/run.err
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('a', 'b'):
    for f in glob.glob('gpurun_out/r6/ros23_pmc_%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'tsit5' not in k and 'forward' not in k and 'adjoint' not in k: continue
            agg[k[:150]][r['Counter_Name']].append(float(r['Counter_Value']))
out = []
for k, v in sorted(agg.items()):
    out.append(k)
    for c, xs in sorted(v.items()):
        out.append('    %-22s launches %4d  mean %.5g  max %.5g' % (c, len(xs), sum(xs) / len(xs), max(xs)))
open('gpurun_out/r6/ros23_pmc_summary.txt', 'w').write('\n'.join(out) + '\n'); print('\n'.join(out)[-6000:])
PY
rm -rf gpurun_out/r6/ros23_pmc_a/*/ gpurun_out/r6/ros23_pmc_b/*/ 2>/dev/null
tail -3 gpurun_out/r6/ros23_pmc_a/run.err
