#!/bin/bash
# round 6, visit 3: the library-side dense-chain route and the grouped one-launch pass (tests), the wave timeline with hardware ids, the top segment's weight (HIPADJ_WTOP) at the
# shard and at the headline size, the grouped form (G waves per workgroup, LDS first level) against the plain one, then the targeted fault stress
O=gpurun_out/r6
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_fused.py tests/test_gpu_device_loss.py -q -m gpu -p no:cacheprovider -k "routed or dense_chain or grouped or loss_value or virtual" > $O/v3_tests.log 2>&1
tail -n 5 $O/v3_tests.log
T=$PWD/scripts/r6/libhipadj_trace.so
HIPADJ_LIBRARY=$T timeout 200 python scripts/r6/wave_trace.py 1250 > $O/v3_wave_1250.jsonl 2> $O/v3_wave.err
HIPADJ_LIBRARY=$T timeout 200 python scripts/r6/wave_trace.py 10000 > $O/v3_wave_10000.jsonl 2>> $O/v3_wave.err
: > $O/v3_wtop.jsonl
for w in 1.6 1.9 2.2 2.4 2.6 2.9; do
  HIPADJ_WTOP=$w timeout 200 python scripts/r6/shard_time.py wtop_$w 1250 2500 10000 >> $O/v3_wtop.jsonl 2>> $O/v3_wtop.err
done
: > $O/v3_group.jsonl
run() { # label n G segs radix
  HIPADJ_FUSED_GROUP=$3 SHARD_SEGMENTS=$4 HIPADJ_TREE_RADIX=$5 timeout 200 python scripts/r6/shard_time.py "$1" $2 >> $O/v3_group.jsonl 2>> $O/v3_group.err
}
run plain_51 1250 0 0 4
run g4_48_r4 1250 4 48 4; run g4_48_r16 1250 4 48 16; run g4_52_r16 1250 4 52 16; run g8_96_r4 1250 8 96 4; run g8_96_r16 1250 8 96 16; run g8_80_r16 1250 8 80 16; run g8_64_r8 1250 8 64 8
run plain 2500 0 0 4; run g4_24_r8 2500 4 24 8; run g8_48_r8 2500 8 48 8; run g4_48_r16 2500 4 48 16
run plain 5000 0 0 4; run g4_12_r4 5000 4 12 4; run g8_24_r4 5000 8 24 4; run g4_24_r8 5000 4 24 8
run plain 10000 0 0 4; run g4_12_r4 10000 4 12 4; run g4_8_r4 10000 4 8 4; run g8_16_r4 10000 8 16 4
cut -c1-170 $O/v3_group.jsonl
HIPADJ_LIBRARY=$T HIPADJ_FUSED_GROUP=8 HIPADJ_TREE_RADIX=16 timeout 200 python scripts/r6/wave_trace.py 1250 96 > $O/v3_wave_1250_g8.jsonl 2>> $O/v3_wave.err
HIPADJ_LIBRARY=$T HIPADJ_FUSED_GROUP=4 HIPADJ_TREE_RADIX=16 timeout 200 python scripts/r6/wave_trace.py 1250 48 > $O/v3_wave_1250_g4.jsonl 2>> $O/v3_wave.err
HIPADJ_TRACE_PIN=1 timeout 900 python scripts/r6/fault_stress.py 3 > $O/v3_stress.json 2> $O/v3_stress.err
echo "stress rc=$?"; cat $O/v3_stress.json; tail -n 2 $O/v3_stress.err
