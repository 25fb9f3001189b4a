#!/bin/bash
# round 3, GPU visit 19: rows-form GroupNorm apply: kernel tests, A/B on the C2 forward, bench
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v19_round.log
echo "$(date)" > $LOG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/r3v19_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v19_tests.log | head -20 >> $LOG
for MODE in 1 0 1 0; do
  GM_GN_APPLY_ROWS=$MODE timeout 300 python tools/ab_lib.py "gn_apply_rows=$MODE" >> $OUT/r3v19_ab.jsonl 2>> $OUT/r3v19_ab.err
done
cat $OUT/r3v19_ab.jsonl >> $LOG
timeout 600 python bench.py --cpu-baseline off > $OUT/r3v19_bench.json 2> $OUT/r3v19_bench.err
echo "bench rc=$?" >> $LOG; cat $OUT/r3v19_bench.json >> $LOG
echo "done $(date)" >> $LOG
