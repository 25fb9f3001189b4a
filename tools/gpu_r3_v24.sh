#!/bin/bash
# round 3, GPU visit 24: generic convolution kernel with unconditional loads: whole suite, C3 per-kernel durations, C3 / C4 benches
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
LOG=$OUT/r3v24_round.log
echo "$(date)" > $LOG
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -rP -p no:cacheprovider > $OUT/r3v24_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v24_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v24_tests.log > $OUT/r3v24_parity.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v24_prof -o c3 -- python $R/tools/layer_times_c3.py > $R/$OUT/r3v24_prof.log 2>&1)
F=$(find $OUT/r3v24_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/r3v24_c3_kernel_stats.csv
rm -rf $OUT/r3v24_prof
timeout 300 python tools/bench_c3.py > $OUT/r3v24_c3.json 2> $OUT/r3v24_c3.err; grep '^{' $OUT/r3v24_c3.json | head -c 300 >> $LOG
timeout 600 python tools/bench_train.py > $OUT/r3v24_train.json 2> $OUT/r3v24_train.err; grep '^{' $OUT/r3v24_train.json | head -c 1800 >> $LOG
timeout 300 python tools/bench_c1b.py > $OUT/r3v24_c1b.json 2> $OUT/r3v24_c1b.err; grep '^{' $OUT/r3v24_c1b.json | head -c 800 >> $LOG
echo "done $(date)" >> $LOG
