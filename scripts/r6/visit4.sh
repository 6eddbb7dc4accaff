#!/bin/bash
# round 6, visit 4: the reduced reproducer of the read-only-page fault (four variants, each in its own process), the stress run on the library with the staging block in its own
# guarded mapping, the shard sizes with the planner's grouped defaults, the whole GPU suite on four workers, a default bench run
O=gpurun_out/r6
mkdir -p $O
: > $O/v4_repro.log
for v in heap_untouched heap_touched no_register mmap_block; do
  echo "== $v" >> $O/v4_repro.log
  timeout 120 scripts/r6/repro_readonly_fault $v 60 >> $O/v4_repro.log 2>&1
  echo "rc=$?" >> $O/v4_repro.log
done
cat $O/v4_repro.log
HIPADJ_TRACE_PIN=1 timeout 900 python scripts/r6/fault_stress.py 4 > $O/v4_stress.json 2> $O/v4_stress.err
echo "stress rc=$?"; cat $O/v4_stress.json; tail -n 3 $O/v4_stress.err
timeout 300 python scripts/r6/shard_time.py planner_defaults 640 1250 2500 5000 10000 20000 > $O/v4_shard_defaults.jsonl 2> $O/v4_shard_defaults.err
HIPADJ_FUSED_GROUP=0 timeout 300 python scripts/r6/shard_time.py plain_form 640 1250 2500 5000 10000 >> $O/v4_shard_defaults.jsonl 2>> $O/v4_shard_defaults.err
cut -c1-150 $O/v4_shard_defaults.jsonl
( time timeout 1100 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $O/v4_gpu_suite.log 2>&1
tail -n 6 $O/v4_gpu_suite.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/v4_bench.json 2> $O/v4_bench.err; cp bench_extras.json $O/v4_bench_extras.json
cat $O/v4_bench.json | cut -c1-1200
