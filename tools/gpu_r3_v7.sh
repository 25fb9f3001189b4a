#!/bin/bash
# round 3, GPU visit 7: C5 decode step, merges in consumer prologues (attention partials, MLP K-slice partials): parity + timing
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v7_round.log
echo "$(date)" > $LOG
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle_r3.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_models.py -m gpu -q -k "c5 or transformer" --maxfail=10 -rP -p no:cacheprovider > $OUT/r3v7_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v7_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v7_tests.log > $OUT/r3v7_parity.txt
for MODE in "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $MODE
  echo "kv_fuse=$1 mlp_fuse=$2" >> $OUT/r3v7_diag.txt
  GM_DECODE_KV_FUSE=$1 GM_DECODE_MLP_FUSE=$2 timeout 300 python tools/diag_c5.py >> $OUT/r3v7_diag.txt 2>&1
done
cat $OUT/r3v7_diag.txt >> $LOG
timeout 600 python tools/bench_c5.py > $OUT/r3v7_c5.json 2> $OUT/r3v7_c5.err; tail -c 1200 $OUT/r3v7_c5.json >> $LOG
echo "done $(date)" >> $LOG
