#!/bin/bash
# round 3, GPU visit 12: row-batched statistics / split-K combine kernels: whole suite, C3 per-kernel durations, C3 + C4 + C5 benches
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
LOG=$OUT/r3v12_round.log
echo "$(date)" > $LOG
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=5 -rP -p no:cacheprovider > $OUT/r3v12_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v12_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v12_tests.log > $OUT/r3v12_parity.txt
timeout 300 python tools/layer_times_c3.py > $OUT/r3v12_layer_times_c3.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v12_prof -o c3 -- python $R/tools/layer_times_c3.py > $R/$OUT/r3v12_prof.log 2>&1)
F=$(find $OUT/r3v12_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/r3v12_c3_kernel_stats.csv
rm -rf $OUT/r3v12_prof
timeout 300 python tools/bench_c3.py > $OUT/r3v12_c3.json 2> $OUT/r3v12_c3.err; tail -c 600 $OUT/r3v12_c3.json | head -c 600 >> $LOG
timeout 600 python tools/bench_train.py > $OUT/r3v12_train.json 2> $OUT/r3v12_train.err; tail -c 1800 $OUT/r3v12_train.json >> $LOG
echo "done $(date)" >> $LOG
