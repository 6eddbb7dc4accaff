#!/bin/bash
# round 6, visit 6: the reduced C reproducer with the full host sequence (system runtime and the runtime torch bundles), the issue-priority A/B of the two waves of a SIMD, the
# shard sizes with the tuned top-segment weight, fifty consecutive default bench runs on the staged library
O=gpurun_out/r6
mkdir -p $O
: > $O/v6_repro.log
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
for mode in system torch; do
  for v in host_sequence heap_untouched; do
    echo "== $v ($mode runtime)" >> $O/v6_repro.log
    if [ $mode = torch ]; then LD_LIBRARY_PATH=$TL timeout 120 scripts/r6/repro_readonly_fault $v 40 >> $O/v6_repro.log 2>&1; else timeout 120 scripts/r6/repro_readonly_fault $v 40 >> $O/v6_repro.log 2>&1; fi
    echo "rc=$?" >> $O/v6_repro.log
  done
done
cat $O/v6_repro.log
: > $O/v6_prio.jsonl
for rep in 1 2; do
  timeout 200 python scripts/r6/shard_time.py default_$rep 10000 5000 >> $O/v6_prio.jsonl 2>> $O/v6_prio.err
  HIPADJ_LIBRARY=$PWD/scripts/r6/libhipadj_prio0.so timeout 200 python scripts/r6/shard_time.py prio_toggle_every_block_$rep 10000 5000 >> $O/v6_prio.jsonl 2>> $O/v6_prio.err
  HIPADJ_LIBRARY=$PWD/scripts/r6/libhipadj_prio2.so timeout 200 python scripts/r6/shard_time.py prio_toggle_every_4_blocks_$rep 10000 5000 >> $O/v6_prio.jsonl 2>> $O/v6_prio.err
done
cut -c1-150 $O/v6_prio.jsonl
timeout 200 python scripts/r6/shard_time.py tuned_wtop 640 1250 2500 5000 10000 > $O/v6_shard_defaults.jsonl 2> $O/v6_shard.err
cut -c1-150 $O/v6_shard_defaults.jsonl
bash scripts/r6/fault_hunt.sh 50
cp gpurun_out/r6/hunt/summary.jsonl $O/v6_hunt_50_default_runs.jsonl
