#!/bin/bash
# round 6, visit 5: A/B of the three host-transfer modes against the read-only-page fault, the stress run on the staged library, the whole GPU suite, the top segment's weight in the
# grouped form, two RCCL ranks on one device, a default bench run and its rocprofv3 kernel statistics
O=gpurun_out/r6
mkdir -p $O
timeout 1500 python scripts/r6/fault_ab.py 12 > $O/v5_fault_ab.jsonl 2> $O/v5_fault_ab.err
cat $O/v5_fault_ab.jsonl | cut -c1-400
timeout 900 python scripts/r6/fault_stress.py 3 > $O/v5_stress.json 2> $O/v5_stress.err
echo "stress rc=$?"; cat $O/v5_stress.json; grep -v host_pin $O/v5_stress.err | tail -n 3
( time timeout 1100 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $O/v5_gpu_suite.log 2>&1
tail -n 6 $O/v5_gpu_suite.log
: > $O/v5_wtop_grouped.jsonl
for w in 1.8 2.1 2.4 2.6 3.0; do
  HIPADJ_WTOP=$w timeout 200 python scripts/r6/shard_time.py grouped_wtop_$w 1250 2500 10000 >> $O/v5_wtop_grouped.jsonl 2>> $O/v5_wtop.err
done
cut -c1-140 $O/v5_wtop_grouped.jsonl
timeout 300 python scripts/r6/rccl_two_ranks_one_device.py > $O/v5_rccl_two_ranks_one_device.json 2> $O/v5_rccl.err
cut -c1-1200 $O/v5_rccl_two_ranks_one_device.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/v5_bench.json 2> $O/v5_bench.err; cp bench_extras.json $O/v5_bench_extras.json
cut -c1-1500 $O/v5_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/v5_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/$O/v5_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/v5_bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
find $O/v5_prof -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} $O/v5_rocprofv3_kernel_stats.csv
find $O/v5_prof -name "*kernel_trace.csv" | head -n 1 | xargs -I{} sh -c 'grep -c k_interp {} ; head -n 1 {} > '$O'/v5_kernel_trace_kinterp.csv; grep k_interp {} | tail -n 60 >> '$O'/v5_kernel_trace_kinterp.csv'
rm -rf $O/v5_prof
head -n 6 $O/v5_rocprofv3_kernel_stats.csv | cut -c1-250
