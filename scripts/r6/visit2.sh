#!/bin/bash
# round 6, visit 2: the GPU suite on four xdist workers (wall time), the wave timeline of a 1250-trajectory shard (trace builds: kernarg-inline vs global segment bounds), the shard sizes
# with both libraries, then the hunt for round 5's memory fault (consecutive default bench runs)
O=gpurun_out/r6
mkdir -p $O
( time timeout 1100 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $O/v2_gpu_suite_xdist.log 2>&1
tail -n 4 $O/v2_gpu_suite_xdist.log
for v in trace trace_gb; do
  HIPADJ_LIBRARY=$PWD/scripts/r6/libhipadj_$v.so timeout 200 python scripts/r6/wave_trace.py 1250 > $O/v2_wave_$v.jsonl 2> $O/v2_wave_$v.err
done
HIPADJ_LIBRARY=$PWD/scripts/r6/libhipadj_trace.so timeout 200 python scripts/r6/wave_trace.py 10000 > $O/v2_wave_trace_10000.jsonl 2>> $O/v2_wave_trace.err
timeout 200 python scripts/r6/shard_time.py inline_bounds > $O/v2_shard_time.jsonl 2> $O/v2_shard_time.err
HIPADJ_LIBRARY=$PWD/scripts/r6/libhipadj_gb.so timeout 200 python scripts/r6/shard_time.py global_bounds >> $O/v2_shard_time.jsonl 2>> $O/v2_shard_time.err
timeout 200 python scripts/r6/shard_time.py inline_bounds_again 1250 2500 >> $O/v2_shard_time.jsonl 2>> $O/v2_shard_time.err
cat $O/v2_shard_time.jsonl | cut -c1-200
bash scripts/r6/fault_hunt.sh ${HUNT_RUNS:-40}
