#!/bin/bash
# round 3, final GPU visit: whole suite, smoke, the default bench run (CPU baseline included), C3 / C4 / C5 / C1b benches on the final tree
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3final_round.log
echo "$(date)" > $LOG
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 -rP -p no:cacheprovider > $OUT/r3final_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3final_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3final_tests.log > $OUT/r3final_parity.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r3final_smoke.log 2>&1; tail -2 $OUT/r3final_smoke.log >> $LOG
timeout 900 python bench.py > $OUT/r3final_bench.json 2> $OUT/r3final_bench.err
echo "bench rc=$?" >> $LOG; grep '^{' $OUT/r3final_bench.json >> $LOG
timeout 300 python tools/bench_c3.py > $OUT/r3final_c3.json 2> $OUT/r3final_c3.err; grep '^{' $OUT/r3final_c3.json | head -c 300 >> $LOG; echo >> $LOG
timeout 600 python tools/bench_train.py > $OUT/r3final_train.json 2> $OUT/r3final_train.err; grep '^{' $OUT/r3final_train.json | head -c 900 >> $LOG; echo >> $LOG
timeout 300 python tools/bench_c5.py > $OUT/r3final_c5.json 2> $OUT/r3final_c5.err; grep '^{' $OUT/r3final_c5.json | head -c 400 >> $LOG; echo >> $LOG
timeout 300 python tools/bench_c1b.py > $OUT/r3final_c1b.json 2> $OUT/r3final_c1b.err; grep '^{' $OUT/r3final_c1b.json | head -c 600 >> $LOG; echo >> $LOG
echo "done $(date)" >> $LOG
