#!/bin/bash
# round 3, GPU visit 4: the activation-vector fix (conv_act_vec) -- whole suite, prologue-placement A/B on the C2 forward, out-head timing, bench
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v4_round.log
echo "$(date)" > $LOG
for MODE in "auto 0" "always 0" "always 1" "never 0" "auto 0" "always 0" "always 1"; do
  set -- $MODE
  GM_DMA_FUSED_PROLOGUE=$1 GM_CONV_WIDE_WAVES_PRE=$2 timeout 300 python tools/ab_lib.py "prologue=$1,wide_pre=$2" >> $OUT/r3v4_ab.jsonl 2>> $OUT/r3v4_ab.err
done
cat $OUT/r3v4_ab.jsonl >> $LOG
timeout 300 python tools/diag_cout1.py > $OUT/r3v4_cout1_time.txt 2>&1
cat $OUT/r3v4_cout1_time.txt >> $LOG
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 -rP -p no:cacheprovider > $OUT/r3v4_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v4_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v4_tests.log > $OUT/r3v4_parity.txt
timeout 600 python bench.py --cpu-baseline off > $OUT/r3v4_bench.json 2> $OUT/r3v4_bench.err
echo "bench rc=$?" >> $LOG; cat $OUT/r3v4_bench.json >> $LOG
GM_DMA_FUSED_PROLOGUE=always timeout 600 python bench.py --cpu-baseline off > $OUT/r3v4_bench_always.json 2> $OUT/r3v4_bench_always.err
echo "bench(always) rc=$?" >> $LOG; cat $OUT/r3v4_bench_always.json >> $LOG
GM_DMA_FUSED_PROLOGUE=always timeout 300 python tools/layer_times.py > $OUT/r3v4_layer_times_always.txt 2>&1
timeout 300 python tools/bench_c3.py > $OUT/r3v4_c3.json 2> $OUT/r3v4_c3.err; tail -c 1500 $OUT/r3v4_c3.json >> $LOG
timeout 300 python tools/bench_c1b.py > $OUT/r3v4_c1b.json 2> $OUT/r3v4_c1b.err; tail -c 800 $OUT/r3v4_c1b.json >> $LOG
echo "done $(date)" >> $LOG
